// Host-only pieces of the C ABI: version, thread-local error text, padding / length rules.
#include "common.h"

#include <cstring>
#include <cstdlib>
#include <mutex>
#include <vector>

namespace nbasr {

static thread_local char g_error[512] = {0};

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}

void clear_error() { g_error[0] = '\0'; }

// Round 5: a chain of `frames` dependent launches costs the host 0.6-0.7 ms per forward to issue -- at 8 utterances per GPU half of the
// step, on the thread that also feeds the encoder's stream; a graph launch costs ~15 us.  Same kernels, same order, same arguments:
// bit-identical results.  (same-box A/B, alternating, bench.py --batch 8 / 16 / 64: 4 856 / 4 781 -> 5 036 / 4 873, 7 172 / 7 048 ->
// 7 258 / 7 248, 9 821 / 9 839 -> 9 852 / 9 896.)  Round 6: graphs only for recurring keys, per-device caches, launches
// under the lock, deferred destruction -- see common.h.
int replay_chain(hipStream_t s, const ChainKey& key, const char* what, void (*launch_chain)(void*), void* ctx)
{
    constexpr int MAX_DEVICES = 64;
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    int device = -1;
    static const bool no_graph = [] { const char* e = getenv("NBASR_CHAIN_GRAPH"); return e && e[0] == '0'; }();   // A/B and diagnosis: always the plain chain
    if (no_graph || hipStreamIsCapturing(s, &capturing) != hipSuccess || capturing != hipStreamCaptureStatusNone ||
        hipGetDevice(&device) != hipSuccess || device < 0 || device >= MAX_DEVICES) {
        (void)hipGetLastError();
        launch_chain(ctx);
        return launch_status(what);
    }
    struct Chain { ChainKey key; hipGraphExec_t exec; hipEvent_t last; unsigned long long used; };
    struct Seen { ChainKey key; int count; unsigned long long used; };
    struct PerDevice { std::mutex m; std::vector<Chain> cache; std::vector<Seen> seen; unsigned long long tick = 0; };
    static PerDevice per_device[MAX_DEVICES];
    constexpr size_t CHAIN_CACHE = 16, SEEN_TABLE = 64;        // per device: 2 pipelined slots x a few (batch, frames) shapes
    constexpr int SEEN_BEFORE_CAPTURE = 3;
    PerDevice& pd = per_device[device];
    std::lock_guard<std::mutex> lock(pd.m);                    // (held across the launch: an entry cannot be destroyed under a launcher)
    Chain* hit = nullptr;
    for (Chain& e : pd.cache) if (e.key == key) { hit = &e; break; }
    if (hit == nullptr) {
        Seen* sn = nullptr;
        for (Seen& e : pd.seen) if (e.key == key) { sn = &e; break; }
        if (sn == nullptr) {
            if (pd.seen.size() >= SEEN_TABLE) {
                size_t lru = 0;
                for (size_t i = 1; i < pd.seen.size(); ++i) if (pd.seen[i].used < pd.seen[lru].used) lru = i;
                pd.seen.erase(pd.seen.begin() + lru);
            }
            pd.seen.push_back(Seen{key, 0, 0});
            sn = &pd.seen.back();
        }
        sn->used = ++pd.tick;
        if (++sn->count < SEEN_BEFORE_CAPTURE) {                // not (yet) a recurring call: the plain chain
            launch_chain(ctx);
            return launch_status(what);
        }
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        hipEvent_t last = nullptr;
        hipError_t e = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
        if (e == hipSuccess) {
            launch_chain(ctx);
            e = hipStreamEndCapture(s, &graph);
        }
        if (e == hipSuccess) e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        if (graph) (void)hipGraphDestroy(graph);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&last, hipEventDisableTiming);
        if (e != hipSuccess || exec == nullptr) {                 // no graph on this runtime / stream: the plain chain
            (void)hipGetLastError();
            if (exec) (void)hipGraphExecDestroy(exec);
            sn->count = 0;
            launch_chain(ctx);
            return launch_status(what);
        }
        if (pd.cache.size() >= CHAIN_CACHE) {                      // the least recently used one goes -- once its last launch has completed
            size_t lru = 0;
            for (size_t i = 1; i < pd.cache.size(); ++i) if (pd.cache[i].used < pd.cache[lru].used) lru = i;
            (void)hipEventSynchronize(pd.cache[lru].last);
            (void)hipGraphExecDestroy(pd.cache[lru].exec);
            (void)hipEventDestroy(pd.cache[lru].last);
            pd.cache.erase(pd.cache.begin() + lru);
        }
        pd.cache.push_back(Chain{key, exec, last, 0});
        hit = &pd.cache.back();
    }
    hit->used = ++pd.tick;
    hipError_t e = hipGraphLaunch(hit->exec, s);
    if (e == hipSuccess) e = hipEventRecord(hit->last, s);
    if (e != hipSuccess) { set_error("%s: hipGraphLaunch: %s", what, hipGetErrorString(e)); return static_cast<int>(e); }
    return launch_status(what);
}

__global__ __launch_bounds__(256) void zero_words_kernel(unsigned* __restrict__ p, size_t words)
{
    for (size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x; i < words; i += static_cast<size_t>(gridDim.x) * blockDim.x) p[i] = 0u;
}
void zero_async(void* p, size_t bytes, hipStream_t stream)
{
    const size_t words = bytes / 4;
    if (words == 0) return;
    const size_t blocks = (words + 1023) / 1024;
    hipLaunchKernelGGL(zero_words_kernel, dim3(static_cast<unsigned>(blocks < 256 ? blocks : 256)), dim3(256), 0, stream, static_cast<unsigned*>(p), words);
}

}  // namespace nbasr

extern "C" int nbasr_version(void) { return NBASR_ABI_VERSION; }

#ifndef NBASR_BUILD_ID
#define NBASR_BUILD_ID "unknown"
#endif
extern "C" const char* nbasr_build_id(void) { return NBASR_BUILD_ID; }

extern "C" const char* nbasr_last_error(void) { return nbasr::g_error; }

extern "C" int nbasr_pad_amounts(int kernel, int dilation, int stride, int* left, int* right) {
    nbasr::clear_error();
    NBASR_REQUIRE(left && right, NBASR_ENULL, "nbasr_pad_amounts: NULL output pointer");
    NBASR_REQUIRE(kernel >= 1 && dilation >= 1 && stride >= 1, NBASR_EINVAL,
                  "nbasr_pad_amounts: kernel=%d dilation=%d stride=%d must all be >= 1", kernel, dilation, stride);
    *left = nbasr::pad_left(kernel, dilation, stride);
    *right = nbasr::pad_right(kernel, dilation, stride);
    return NBASR_OK;
}

extern "C" int nbasr_stream_create(nbasr_stream_t* stream) {
    nbasr::clear_error();
    NBASR_REQUIRE(stream, NBASR_ENULL, "nbasr_stream_create: NULL output pointer");
    hipStream_t s = nullptr;
    const hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    if (e != hipSuccess) { (void)hipGetLastError(); nbasr::set_error("nbasr_stream_create: %s", hipGetErrorString(e)); return static_cast<int>(e); }
    *stream = s;
    return NBASR_OK;
}

extern "C" int nbasr_output_frames(int frames) {
    if (frames <= 0) return 0;
    const int half = (frames + 1) / 2;
    return (half + 1) / 2;
}
