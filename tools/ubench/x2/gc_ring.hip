// Exploration (round 3): the fp32 node kernel with its input windows staged through LDS by LDS-DMA (buffer_load ... lds).
//
// Reading of round 2's measurements: the node kernel is bound by the bytes a CU keeps in flight (a wave of the library kernel has
// 1-2 KiB of DISTINCT bytes outstanding, 24-32 KiB per CU, where 6.3 TB/s x ~2 us of loaded latency wants ~49 KiB), and every way of
// deepening the lookahead through registers cost occupancy.  LDS-DMA needs no registers: here a wave requests ALL CG input rows of
// its tile up front (CG KiB per wave, 72-144 KiB per CU), consumes them channel by channel behind counted vmcnt waits, and -- in the
// persistent form -- refills a channel's slot with the NEXT tile's row as soon as that slot has been read into registers.
//
//   slot (per wave, per input channel): [64 main quads = 1 KiB][QL left + QR right halo quads, padded to 64 B]
//   main DMA : lane l <- quad q0 + l of the row (beyond the row: zeros from the buffer bounds check)
//   halo DMA : lanes 0..QL+QR-1 <- quads q0 - QL .. q0 - 1, q0 + 64 .. q0 + 63 + QR (only when a row is longer than one tile)
//   window   : NCH ds_read_b128 per lane and channel at per-lane offsets computed once
// Same sums in the same order as the library kernel: bit-identical (asserted by the harness).
//
// build: hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -shared -I include -I nb_asr_amd/csrc -x hip tools/ubench/x2/gc_ring.hip -o libgc_ring.so
#include "common.h"

namespace x2 {
using namespace nbasr;

typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int SLOT = 1024 + 64;

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// wait until at most n vector-memory operations of this wave are outstanding (n is wave-uniform)
__device__ __forceinline__ void wait_vm_dyn(int n)
{
    switch (n) {
#define X2_CASE(N) case N: wait_vm<N>(); break;
        X2_CASE(0) X2_CASE(1) X2_CASE(2) X2_CASE(3) X2_CASE(4) X2_CASE(5) X2_CASE(6) X2_CASE(7) X2_CASE(8) X2_CASE(9)
        X2_CASE(10) X2_CASE(11) X2_CASE(12) X2_CASE(13) X2_CASE(14) X2_CASE(15) X2_CASE(16) X2_CASE(17) X2_CASE(18) X2_CASE(19)
        X2_CASE(20) X2_CASE(21) X2_CASE(22) X2_CASE(23) X2_CASE(24) X2_CASE(25) X2_CASE(26) X2_CASE(27) X2_CASE(28) X2_CASE(29)
        X2_CASE(30) X2_CASE(31) X2_CASE(32) X2_CASE(33) X2_CASE(34) X2_CASE(35) X2_CASE(36) X2_CASE(37) X2_CASE(38) X2_CASE(39)
        X2_CASE(40) X2_CASE(41) X2_CASE(42) X2_CASE(43) X2_CASE(44) X2_CASE(45) X2_CASE(46) X2_CASE(47) X2_CASE(48)
#undef X2_CASE
        default: wait_vm<0>(); break;
    }
}

// work item of a workgroup = (utterance, quad of groups, 64-quad frame tile); the four waves take the quad's four groups
template <int CG, int K, int D, bool HALO, bool PERSIST>
__global__ __launch_bounds__(256) void gc_ring_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                      float* __restrict__ y, int channels, int frames, int ld, int groups, int batch,
                                                      int n_xt, int n_items)
{
    constexpr int LPAD = pad_left(K, D, 1);
    constexpr int SPAN = (K - 1) * D;
    constexpr int QL = (LPAD + 3) / 4;
    constexpr int QR = (SPAN - LPAD + 3) / 4;
    constexpr int NCH = QL + 1 + QR;
    constexpr int BASE = 4 * QL - LPAD;
    constexpr int H = QL + QR;
    constexpr int DPC = HALO ? 2 : 1;                 // DMA instructions per channel
    static_assert(H * 16 <= 64, "halo area");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned char* const ring = smem + wave * (CG * SLOT);
    const int nq = ld >> 2, row_bytes = ld * 4;
    const int n_gq = (groups + 3) >> 2;

    // per-lane read offsets of the NCH window chunks inside a slot
    int rd[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int r = lane - QL + c;
        rd[c] = (r >= 0 && r < 64) ? r * 16 : (r < 0 ? 1024 + (r + QL) * 16 : 1024 + (QL + r - 64) * 16);
    }
    if (!HALO) {
        // rows no longer than one tile: everything outside the tile is zero padding; the halo areas are written once, here
        if (lane < 4 * CG) *reinterpret_cast<f4*>(ring + (lane >> 2) * SLOT + 1024 + (lane & 3) * 16) = f4{0.f, 0.f, 0.f, 0.f};
    }

    auto decode = [&](int item, int& b, int& g, int& q0) {
        const int xt = item % n_xt;
        const int rest = item / n_xt;
        const int gq = rest % n_gq;
        b = rest / n_gq;
        g = gq * 4 + wave;
        q0 = xt * 64;
    };
    // request the CG input rows of a tile (channel ci -> slot ci)
    auto issue_channel = [&](const float* xg, int ci, int q0) {
        const float* row = xg + static_cast<size_t>(ci) * ld;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(row), 0, row_bytes, 0x00020000);
        unsigned char* slot = ring + ci * SLOT;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)slot, 16, (q0 + lane) * 16, 0, 0, 0);
        if (HALO) {
            const int hq = lane < QL ? q0 - QL + lane : q0 + 64 + (lane - QL);
            if (lane < H)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(slot + 1024), 16, hq * 16, 0, 0, 0);
        }
    };

    int item = blockIdx.x;
    if (item >= n_items) return;
    int b, g, q0;
    decode(item, b, g, q0);
    if (g >= groups) return;                          // (a surplus wave of the last quad: it takes no part at all)
    const float* xg = x + (static_cast<size_t>(b) * channels + static_cast<size_t>(g) * CG) * ld;
#pragma unroll 1
    for (int ci = 0; ci < CG; ++ci) issue_channel(xg, ci, q0);

    bool have_prev = false;
    while (true) {
        const int next = PERSIST ? item + static_cast<int>(gridDim.x) : n_items;
        const bool have_next = next < n_items;
        int nb = b, ng = g, nq0 = q0;
        if (have_next) decode(next, nb, ng, nq0);
        const float* nxg = x + (static_cast<size_t>(nb) * channels + static_cast<size_t>(ng) * CG) * ld;
        const float* __restrict__ wg = w + static_cast<size_t>(g) * (CG * CG * K);
        const float* __restrict__ bg = bias + g * CG;

        float acc[CG][4];
#pragma unroll
        for (int co = 0; co < CG; ++co) {
            const float bv = bg[co];
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[co][r] = bv;
        }
#pragma unroll 1
        for (int ci = 0; ci < CG; ++ci) {
            // operations younger than this channel's DMAs: the rest of this tile's, the previous tile's CG stores, the next tile's so far
            wait_vm_dyn((CG - 1 - ci) * DPC + (have_prev ? CG : 0) + (have_next ? ci * DPC : 0));
            const unsigned char* slot = ring + ci * SLOT;
            float xw[NCH * 4];
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const f4 v = *reinterpret_cast<const f4*>(slot + rd[c]);
                xw[4 * c + 0] = v[0]; xw[4 * c + 1] = v[1]; xw[4 * c + 2] = v[2]; xw[4 * c + 3] = v[3];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the window is in registers: the slot may be refilled
            if (have_next) issue_channel(nxg, ci, nq0);
#pragma unroll
            for (int j = 0; j < K; ++j) {
#pragma unroll
                for (int co = 0; co < CG; ++co) {
                    const float wv = wg[(co * CG + ci) * K + j];
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[co][r] = __builtin_fmaf(wv, xw[BASE + r + j * D], acc[co][r]);
                }
            }
        }
        // epilogue: bias + ReLU + clamp (+ zeroed pitch columns), CG bounds-checked buffer stores (exactly CG per wave: they count in vmcnt)
        const int t0 = (q0 + lane) * 4;
        float* yg = y + (static_cast<size_t>(b) * channels + static_cast<size_t>(g) * CG) * ld;
#pragma unroll
        for (int co = 0; co < CG; ++co) {
            f4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (t0 + r < frames) ? relu_clamp(acc[co][r]) : 0.f;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(yg + static_cast<size_t>(co) * ld, 0, row_bytes, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, o), rs, (q0 + lane) * 16, 0, 2);
        }
        if (!have_next) break;
        item = next; b = nb; g = ng; q0 = nq0; xg = nxg;
        have_prev = true;
    }
}

template <int CG, int K, int D>
static int launch(int mode, const float* x, const float* w, const float* bias, float* y, int batch, int channels, int frames, int ld,
                  int wgs_per_cu, hipStream_t s)
{
    const int groups = 100, nq = ld / 4, n_xt = (nq + 63) / 64, n_gq = (groups + 3) / 4;
    const int n_items = n_xt * n_gq * batch;
    const size_t lds = 4 * CG * SLOT;
    const bool halo = n_xt > 1;
    const bool persist = mode == 1;
    int grid = n_items;
    if (persist) grid = n_items < 256 * wgs_per_cu ? n_items : 256 * wgs_per_cu;
#define X2_GO(HALO, PERS)                                                                                                              \
    do {                                                                                                                               \
        static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(gc_ring_kernel<CG, K, D, HALO, PERS>),       \
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));        \
        if (attr != hipSuccess) return static_cast<int>(attr);                                                                         \
        hipLaunchKernelGGL((gc_ring_kernel<CG, K, D, HALO, PERS>), dim3(grid), dim3(256), lds, s, x, w, bias, y, channels, frames, ld, \
                           groups, batch, n_xt, n_items);                                                                              \
    } while (0)
    if (halo && persist) X2_GO(true, true);
    else if (halo) X2_GO(true, false);
    else if (persist) X2_GO(false, true);
    else X2_GO(false, false);
#undef X2_GO
    return static_cast<int>(hipGetLastError());
}

}  // namespace x2

// mode: 0 = one tile per wave (grid = tiles), 1 = persistent (grid = wgs_per_cu x 256, a slot is refilled with the next tile's row)
extern "C" int x2_node(int mode, const float* x, const float* w, const float* bias, float* y, int batch, int channels, int frames, int ld,
                       int kernel, int dilation, int wgs_per_cu, void* stream)
{
    hipStream_t s = static_cast<hipStream_t>(stream);
#define X2_KD(CG)                                                                                                          \
    if (kernel == 5 && dilation == 1) return x2::launch<CG, 5, 1>(mode, x, w, bias, y, batch, channels, frames, ld, wgs_per_cu, s); \
    if (kernel == 7 && dilation == 2) return x2::launch<CG, 7, 2>(mode, x, w, bias, y, batch, channels, frames, ld, wgs_per_cu, s); \
    return -2;
    switch (channels / 100) {
        case 6: X2_KD(6)
        case 8: X2_KD(8)
        case 10: X2_KD(10)
        case 12: X2_KD(12)
    }
#undef X2_KD
    return -1;
}
