// Shared helpers for the gfx950 kernels of libnbasr_hip.so (internal; the public ABI is include/nbasr.h).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "nbasr.h"

namespace nbasr {

// thread-local error text behind nbasr_last_error()
void set_error(const char* fmt, ...);
void clear_error();

inline hipStream_t as_stream(nbasr_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

inline int launch_status(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return static_cast<int>(e);
    }
    return NBASR_OK;
}

// host + device copy of the PadConvRelu padding rule (reference ops.py:12-17, context = 4)
__host__ __device__ constexpr int pad_left(int kernel, int dilation, int stride) {
    return (4 / stride >= kernel * dilation - stride) ? 0 : (kernel - 1) * dilation - 4 / stride;
}
__host__ __device__ constexpr int pad_right(int kernel, int dilation, int stride) {
    return (4 / stride >= kernel * dilation - stride) ? kernel * dilation - stride : 4 / stride;
}

constexpr float kClamp = 20.0f;   // reference ops.py:28

// device-side view of nbasr_deferred_ln (stats == nullptr: no pending LayerNorm)
struct LnRef {
    const float* stats; const float* gamma; const float* beta;
};
inline LnRef ln_ref(const nbasr_deferred_ln* ln, bool wanted) {
    return (ln && wanted) ? LnRef{ln->stats, ln->gamma, ln->beta} : LnRef{nullptr, nullptr, nullptr};
}
// xn = (x - mean) * rstd * gamma + beta, exactly 0 where rstd == 0 (pitch columns / frames outside the utterance)
__device__ __forceinline__ float ln_apply(float x, float mean, float rstd, float gamma, float beta) {
    return rstd != 0.f ? __builtin_fmaf((x - mean) * rstd, gamma, beta) : 0.f;
}

// min(relu(v), 20) with the reference's treatment of non-finite values (ops.py:27-28: relu, then clamp_max_): NaN stays NaN,
// +Inf -> 20, -Inf -> 0.  gfx950 has the IEEE-754-2019 minimum / maximum (v_maximum3_f32 / v_minimum3_f32: NaN-propagating), so this
// is two instructions -- as many as rounds 1-3's v_med3 + fma(v, 0, .), which turned +-Inf into NaN (v_med3 alone turns NaN into 0,
// v_max / v_min return the other operand).
__device__ __forceinline__ float relu_clamp(float v) {
    return __builtin_elementwise_minimum(__builtin_elementwise_maximum(v, 0.0f), kClamp);
}
// |v| if it is finite, else 0: range statistics (max|x| per utterance) must not be hijacked by an Inf / NaN sample --
// the scaled fp16 GEMM would otherwise flush every finite sample of that utterance to zero
__device__ __forceinline__ float finite_abs(float v) { const float a = fabsf(v); return a <= 3.4028234664e38f ? a : 0.f; }

// A chain of dependent launches replayed as ONE instantiated graph per key (api.cpp).  `launch_chain` enqueues the chain on `stream`;
// `key` names everything the launches depend on (buffers, shapes, a tag for the entry point).  A graph is only built for a key that has
// been SEEN three times (a caller with fresh buffers or a new frame count per call would otherwise pay capture + instantiate + destroy
// every time and never replay); caches are per device; an entry is launched under the lock and destroyed only after the event behind
// its last launch has completed; a stream that is being captured by the caller takes the plain launches.
struct ChainKey {
    const void* ptr[5];
    int val[5];
    bool operator==(const ChainKey& o) const {
        for (int i = 0; i < 5; ++i) if (ptr[i] != o.ptr[i] || val[i] != o.val[i]) return false;
        return true;
    }
};
int replay_chain(hipStream_t stream, const ChainKey& key, const char* what, void (*launch_chain)(void*), void* ctx);

// Zero `bytes` (a multiple of 4) at `p`, stream-ordered, as a KERNEL.  Not hipMemsetAsync (round 6): captured into a graph -- the cached
// per-frame recurrence chain, a whole-forward graph -- the fill becomes a memset node, and such nodes ran out of order under load: with
// three chains of forwards in flight on three streams of the device, one of them the default stream, NaN logits in every second
// forward (tools/ubench/in_flight_check.py --overlap-main; clean with the plain chain, NBASR_CHAIN_GRAPH=0, and with this kernel node).
void zero_async(void* p, size_t bytes, hipStream_t stream);

#define NBASR_REQUIRE(cond, code, ...)          \
    do {                                        \
        if (!(cond)) {                          \
            ::nbasr::set_error(__VA_ARGS__);    \
            return (code);                      \
        }                                       \
    } while (0)

}  // namespace nbasr
