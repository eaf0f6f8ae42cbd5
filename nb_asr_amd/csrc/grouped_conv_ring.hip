// LDS-ring form of the fp32 node kernel (variant bit NBASR_GC_RING): the input windows of a tile are staged through LDS by LDS-DMA
// (buffer_load ... lds) instead of through registers.
//
// Why (round 3).  Round 2 established that the node kernel is bound by the bytes a CU keeps in flight, not by issue, clock or its
// surroundings: a wave of the default kernel has 1 KiB of distinct input bytes outstanding, the pipelined one 2 KiB, i.e.
// 24-32 KiB per CU where 6.3 TB/s x ~2 us of loaded latency wants ~49 KiB -- and every way of deepening the lookahead through
// REGISTERS (two windows in flight, cooperative loads) paid for it in occupancy.  LDS-DMA needs no registers:
//   * a wave requests ALL CG input rows of its tile up front (CG KiB per wave, 70-150 KiB per CU at 3-6 waves per SIMD) and
//     consumes them channel by channel behind COUNTED vmcnt waits (the first channel's FMAs start when the first row has landed);
//   * (measured and dropped: a PERSISTENT form whose workgroups walk a list of tiles and refill a channel's slot with the next
//     tile's row as soon as it has been read -- 3-13 % slower than one tile per wave in the prototype, tools/ubench/x2, and its
//     dynamic wait counts cost 30-60 registers;)
//   * no output split: with the window in LDS the registers are 4 CG accumulators + one window, 42-73 in all.
// slot (per wave, per input channel): [64 main quads = 1 KiB][QL left + QR right halo quads, padded to 64 B]
//   main DMA: lane l <- quad q0 + l of the row (beyond the row: zeros from the buffer bounds check -- the convolution's padding)
//   halo DMA: lanes 0 .. QL + QR - 1 <- quads q0 - QL .. q0 - 1, q0 + 64 .. q0 + 63 + QR (only when a row is longer than one tile)
//   window  : NCH ds_read_b128 per lane and channel at per-lane offsets computed once
// vmcnt is in issue order for loads, stores and LDS-DMA alike (MI355X_MICROARCH.md): "the DMA of channel ci has landed" = "at most
// N younger operations are outstanding", N = the exact number of vector-memory instructions this wave has issued since -- every
// such instruction below is therefore issued unconditionally (bounds-checked buffer accesses instead of predicates) and counted.
// hipcc neither tracks LDS-DMA -> ds_read dependencies nor disturbs explicit waits; its own counted waits for ordinary loads include
// the DMA instructions (checked in the ISA).
// Every output is the same sum in the same order as in the default kernel: bit-identical (tests/test_bf16_ops_gpu.py).
#include "grouped_conv_impl.h"

#include <type_traits>

namespace nbasr {

typedef float rg_f4 __attribute__((ext_vector_type(4)));
typedef unsigned rg_u4 __attribute__((ext_vector_type(4)));
constexpr int RING_SLOT = 1024 + 64;

template <int N> __device__ __forceinline__ void ring_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// wait until at most n vector-memory operations of this wave are outstanding (n is wave-uniform; vmcnt has 6 bits)
__device__ __forceinline__ void ring_wait_dyn(int n)
{
    switch (n) {
#define NBASR_RW(N) case N: ring_wait<N>(); break;
        NBASR_RW(0) NBASR_RW(1) NBASR_RW(2) NBASR_RW(3) NBASR_RW(4) NBASR_RW(5) NBASR_RW(6) NBASR_RW(7) NBASR_RW(8) NBASR_RW(9)
        NBASR_RW(10) NBASR_RW(11) NBASR_RW(12) NBASR_RW(13) NBASR_RW(14) NBASR_RW(15) NBASR_RW(16) NBASR_RW(17) NBASR_RW(18) NBASR_RW(19)
        NBASR_RW(20) NBASR_RW(21) NBASR_RW(22) NBASR_RW(23) NBASR_RW(24) NBASR_RW(25) NBASR_RW(26) NBASR_RW(27) NBASR_RW(28) NBASR_RW(29)
        NBASR_RW(30) NBASR_RW(31) NBASR_RW(32) NBASR_RW(33) NBASR_RW(34) NBASR_RW(35) NBASR_RW(36) NBASR_RW(37) NBASR_RW(38) NBASR_RW(39)
        NBASR_RW(40) NBASR_RW(41) NBASR_RW(42) NBASR_RW(43) NBASR_RW(44) NBASR_RW(45) NBASR_RW(46) NBASR_RW(47) NBASR_RW(48) NBASR_RW(49)
        NBASR_RW(50) NBASR_RW(51) NBASR_RW(52) NBASR_RW(53) NBASR_RW(54) NBASR_RW(55) NBASR_RW(56) NBASR_RW(57) NBASR_RW(58) NBASR_RW(59)
        NBASR_RW(60) NBASR_RW(61) NBASR_RW(62)
#undef NBASR_RW
        default: ring_wait<0>(); break;        // more than the counter holds (never with CG <= 12): wait for everything
    }
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t ring_rsrc(const float* base, int bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, bytes, 0x00020000);      // raw buffer, bounds-checked
}
__device__ __forceinline__ float4 ring_load4(const float* row, int row_bytes, int byte_offset)
{
    const rg_f4 f = __builtin_bit_cast(rg_f4, __builtin_amdgcn_raw_buffer_load_b128(ring_rsrc(row, row_bytes), byte_offset, 0, 0));
    return make_float4(f[0], f[1], f[2], f[3]);
}

// sizes only: every POINTER is a __restrict__ kernel argument of its own.  The explicit waits below are memory clobbers; hipcc keeps
// wave-uniform loads (weights, bias, gamma / beta) on the scalar path across them only for noalias read-only kernel arguments.  A
// pointer taken from a by-value struct has no such attribute: the first version loaded the weights through the VECTOR memory path --
// which also counts in vmcnt -- and ran 35-70 % slower than the same loop with plain arguments (profiles/r03_ab_gc_ring_first.jsonl).
struct RingDims {
    int channels, frames, ld, groups, batch;
    int n_xt, n_items, halo;
};

// workgroups per CU that the LDS admits (4 waves x CG slots each) = waves per SIMD: the register budget the kernel is compiled for
// (capped at 5, 4 with LayerNorm on load: its packed window statistics take 12 NCH registers)
constexpr int ring_waves_per_simd(int cg, bool lnx)
{
    const int lds = (160 * 1024) / (4 * cg * RING_SLOT), cap = lnx ? 4 : 5;
    return lds > cap ? cap : lds;
}

// work item of a workgroup = (utterance, quad of groups, 64-quad frame tile); its four waves take the quad's four groups
template <int CG, int K, int D, bool LNX, bool STATS>
__global__ __launch_bounds__(256, ring_waves_per_simd(CG, LNX)) void grouped_conv_f32_ring_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
    const float* __restrict__ s0, const float* __restrict__ s1, const float* __restrict__ s2, float* __restrict__ y,
    const float* __restrict__ lnx_stats, const float* __restrict__ lnx_gamma, const float* __restrict__ lnx_beta,
    const float* __restrict__ ln0_stats, const float* __restrict__ ln0_gamma, const float* __restrict__ ln0_beta,
    float* __restrict__ part, const RingDims a)
{
    constexpr int LPAD = pad_left(K, D, 1);
    constexpr int SPAN = (K - 1) * D;
    constexpr int QL = (LPAD + 3) / 4;
    constexpr int QR = (SPAN - LPAD + 3) / 4;
    constexpr int NCH = QL + 1 + QR;
    constexpr int BASE = 4 * QL - LPAD;
    constexpr int H = QL + QR;
    static_assert(H * 16 <= 64, "halo area");
    extern __shared__ __attribute__((aligned(16))) unsigned char ring_smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned char* const ring = ring_smem + wave * (CG * RING_SLOT);
    const int ld = a.ld, nq = ld >> 2, row_bytes = ld * 4, channels = a.channels, groups = a.groups;
    const int n_gq = (groups + 3) >> 2;
    const int dpc = a.halo ? 2 : 1;                   // DMA instructions per channel (wave-uniform)
    const int nsk = (s0 ? 1 : 0) + (s1 ? 1 : 0) + (s2 ? 1 : 0);
    const bool ln0 = s0 && ln0_stats;

    // per-lane read offsets of the NCH window chunks inside a slot
    int rd[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int r = lane - QL + c;
        rd[c] = (r >= 0 && r < 64) ? r * 16 : (r < 0 ? 1024 + (r + QL) * 16 : 1024 + (QL + r - 64) * 16);
    }
    if (!a.halo) {
        // rows no longer than one tile: everything outside the tile is zero padding; the halo areas are written once, here
        if (lane < 4 * CG) *reinterpret_cast<rg_f4*>(ring + (lane >> 2) * RING_SLOT + 1024 + (lane & 3) * 16) = rg_f4{0.f, 0.f, 0.f, 0.f};
    }

    auto decode = [&](int item, int& b, int& g_raw, int& q0) {
        const int xt = item % a.n_xt;
        const int rest = item / a.n_xt;
        const int gq = rest % n_gq;
        b = rest / n_gq;
        g_raw = gq * 4 + wave;
        q0 = xt * 64;
    };
    auto issue_channel = [&](const float* xg, int ci, int q0) {
        const __amdgpu_buffer_rsrc_t rs = ring_rsrc(xg + static_cast<size_t>(ci) * ld, row_bytes);
        unsigned char* slot = ring + ci * RING_SLOT;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)slot, 16, (q0 + lane) * 16, 0, 0, 0);
        if (a.halo) {
            const int hq = lane < QL ? q0 - QL + lane : q0 + 64 + (lane - QL);
            if (lane < H)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(slot + 1024), 16, hq * 16, 0, 0, 0);
        }
    };

    typedef float f2 __attribute__((ext_vector_type(2)));
    constexpr int NP = LNX ? NCH * 2 : 1;
    f2 nmw[NP], rw[NP], kw[NP];
    // LayerNorm statistics of the window (shared by all input channels of the tile): 2 NCH bounds-checked loads
    auto load_window_stats = [&](int b, int q0) {
        const float* __restrict__ mrow = lnx_stats + static_cast<size_t>(b) * 2 * ld;
        const int off0 = (q0 + lane - QL) * 16;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const float4 m = ring_load4(mrow, row_bytes, off0 + 16 * c), r = ring_load4(mrow + ld, row_bytes, off0 + 16 * c);
            nmw[(2 * c) % NP] = f2{-m.x, -m.y}; nmw[(2 * c + 1) % NP] = f2{-m.z, -m.w};
            rw[(2 * c) % NP] = f2{r.x, r.y};    rw[(2 * c + 1) % NP] = f2{r.z, r.w};
            kw[(2 * c) % NP] = f2{r.x != 0.f ? 1.f : 0.f, r.y != 0.f ? 1.f : 0.f};
            kw[(2 * c + 1) % NP] = f2{r.z != 0.f ? 1.f : 0.f, r.w != 0.f ? 1.f : 0.f};
        }
    };

    const int item = blockIdx.x;
    int b, g_raw, q0;
    decode(item, b, g_raw, q0);
    if (!STATS && g_raw >= groups) return;            // a surplus wave of the last quad takes no part (with STATS it must reach the barriers:
    const int g = g_raw < groups ? g_raw : groups - 1;      //  it recomputes the last group and its stores are dropped by an empty descriptor)
    const float* xg = x + (static_cast<size_t>(b) * channels + static_cast<size_t>(g) * CG) * ld;
    if (LNX) load_window_stats(b, q0);                // OLDER than the DMAs: waiting for a row does not wait for them, and vice versa
    asm volatile("" ::: "memory");                    // (pins the order of issue the wait counts assume; no instruction)
#pragma unroll 1
    for (int ci = 0; ci < CG; ++ci) issue_channel(xg, ci, q0);
    asm volatile("" ::: "memory");

    {
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
            ring_wait_dyn((CG - 1 - ci) * dpc);         // operations younger than the DMAs of channel ci: the later channels' DMAs
            const unsigned char* slot = ring + ci * RING_SLOT;
            float xw[NCH * 4];
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const rg_f4 v = *reinterpret_cast<const rg_f4*>(slot + rd[c]);
                xw[4 * c + 0] = v[0]; xw[4 * c + 1] = v[1]; xw[4 * c + 2] = v[2]; xw[4 * c + 3] = v[3];
            }
            if (LNX) {
                const float gam = lnx_gamma[g * CG + ci], bet = lnx_beta[g * CG + ci];
                const f2 gam2 = f2{gam, gam}, bet2 = f2{bet, bet};
#pragma unroll
                for (int p = 0; p < NCH * 2; ++p) {
                    f2 v = f2{xw[2 * p], xw[2 * p + 1]};
                    v = (v + nmw[p % NP]) * rw[p % NP];
                    v = __builtin_elementwise_fma(v, gam2, bet2) * kw[p % NP];
                    xw[2 * p] = v.x; xw[2 * p + 1] = v.y;
                }
            }
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

        asm volatile("" ::: "memory");                // the epilogue's loads and stores are issued after every DMA of the loop above
        // ---- epilogue: ReLU + clamp, skip sum (LayerNorm on load for skip0), zeroed pitch columns, CG stores ----------------------
        const int q = q0 + lane;
        const int t0 = q * 4;
        const int boff = q * 16;
        const size_t row0 = (static_cast<size_t>(b) * channels + static_cast<size_t>(g) * CG) * ld;
        const int store_bytes = (!STATS || g_raw < groups) ? row_bytes : 0;      // a surplus wave's stores are dropped by the bounds check
        // Instantiated per number of skip inputs, branch-free inside (round 2's finding, DESIGN 7.1: behind the wave-uniform
        // `if (skip)` branches of a generic epilogue hipcc waits for vmcnt(0) at every join, so a wave's stores and skip loads wait for
        // one another): the skip loads of four output channels are requested together, then their sums are formed and stored.
        auto epilogue = [&](auto nsk_, auto ln0_) {
            constexpr int NSK = decltype(nsk_)::value;
            constexpr bool LN0 = decltype(ln0_)::value;
            const float* const sk[3] = {s0, s1, s2};
            float4 sm = make_float4(0.f, 0.f, 0.f, 0.f), sr = sm;                // statistics of this lane's own 4 frames (skip0)
            if constexpr (LN0) {
                const float* mrow = ln0_stats + static_cast<size_t>(b) * 2 * ld;
                sm = ring_load4(mrow, row_bytes, boff);
                sr = ring_load4(mrow + ld, row_bytes, boff);
            }
            constexpr int CH = 4;
#pragma unroll
            for (int c0 = 0; c0 < CG; c0 += CH) {
                float4 v[CH][NSK > 0 ? NSK : 1];
#pragma unroll
                for (int c = 0; c < CH; ++c)
#pragma unroll
                    for (int k = 0; k < NSK; ++k)
                        if (c0 + c < CG) v[c][k] = ring_load4(sk[k] + row0 + static_cast<size_t>(c0 + c) * ld, row_bytes, boff);
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    const int co = c0 + c;
                    if (co >= CG) break;
                    float o[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = relu_clamp(acc[co][r]);
#pragma unroll
                    for (int k = 0; k < NSK; ++k) {
                        float4 u = v[c][k];
                        if (LN0 && k == 0) {
                            const float gam = ln0_gamma[g * CG + co], bet = ln0_beta[g * CG + co];
                            u.x = ln_apply(u.x, sm.x, sr.x, gam, bet); u.y = ln_apply(u.y, sm.y, sr.y, gam, bet);
                            u.z = ln_apply(u.z, sm.z, sr.z, gam, bet); u.w = ln_apply(u.w, sm.w, sr.w, gam, bet);
                        }
                        o[0] += u.x; o[1] += u.y; o[2] += u.z; o[3] += u.w;
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (t0 + r >= a.frames) o[r] = 0.f;   // pitch columns stay zero (a select, no branch)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(rg_u4, rg_f4{o[0], o[1], o[2], o[3]}),
                                                           ring_rsrc(y + row0 + static_cast<size_t>(co) * ld, store_bytes), boff, 0, 2);   // aux 2 = nt
                    if (STATS) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[co][r] = o[r];                // keep the final values for the statistics
                    }
                }
            }
        };
        using std::integral_constant;
        switch (nsk * 2 + (ln0 ? 1 : 0)) {
            case 0: epilogue(integral_constant<int, 0>{}, integral_constant<bool, false>{}); break;
            case 2: epilogue(integral_constant<int, 1>{}, integral_constant<bool, false>{}); break;
            case 3: epilogue(integral_constant<int, 1>{}, integral_constant<bool, true>{}); break;
            case 4: epilogue(integral_constant<int, 2>{}, integral_constant<bool, false>{}); break;
            case 5: epilogue(integral_constant<int, 2>{}, integral_constant<bool, true>{}); break;
            case 6: epilogue(integral_constant<int, 3>{}, integral_constant<bool, false>{}); break;
            default: epilogue(integral_constant<int, 3>{}, integral_constant<bool, true>{}); break;
        }
        if constexpr (STATS) {
            // per-lane (mean, M2) over this group's CG channels, exact two-pass in registers; wave 0 merges the workgroup's groups
            // (same arithmetic and order as the default kernel's statistics epilogue)
            __shared__ float sp[4][8][64];
            float pm[4], p2[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float sum = 0.f;
#pragma unroll
                for (int co = 0; co < CG; ++co) sum += acc[co][r];
                pm[r] = sum * (1.0f / CG);
                float m2 = 0.f;
#pragma unroll
                for (int co = 0; co < CG; ++co) { const float d = acc[co][r] - pm[r]; m2 = __builtin_fmaf(d, d, m2); }
                p2[r] = m2;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) { sp[wave][r][lane] = pm[r]; sp[wave][4 + r][lane] = p2[r]; }
            __syncthreads();
            if (wave == 0) {
                const int g0 = (g_raw >> 2) * 4;
                const int nw = min(4, groups - g0);                      // groups (waves) that hold real data
                float om[4], o2[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float mean = 0.f;
                    for (int k = 0; k < nw; ++k) mean += sp[k][r][lane];
                    mean /= static_cast<float>(nw);
                    float m2 = 0.f;
                    for (int k = 0; k < nw; ++k) { const float d = sp[k][r][lane] - mean; m2 += sp[k][4 + r][lane] + CG * d * d; }
                    om[r] = mean; o2[r] = m2;
                }
                float* prow = part + (static_cast<size_t>(g_raw >> 2) * a.batch + b) * 2 * ld;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(rg_u4, rg_f4{om[0], om[1], om[2], om[3]}), ring_rsrc(prow, row_bytes), boff, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(rg_u4, rg_f4{o2[0], o2[1], o2[2], o2[3]}), ring_rsrc(prow + ld, row_bytes), boff, 0, 0);
            }
        }
    }
}

template <int CG, int K, int D, bool LNX, bool STATS>
static int launch_ring(int /* variant */, const GroupedArgs<float>& g, const RingDims& a, hipStream_t stream)
{
    const size_t lds = 4 * CG * RING_SLOT;
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(grouped_conv_f32_ring_kernel<CG, K, D, LNX, STATS>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    if (attr != hipSuccess) { set_error("nbasr_grouped_conv1d_node(ring): cannot reserve %zu bytes of LDS", lds); return static_cast<int>(attr); }
    hipLaunchKernelGGL((grouped_conv_f32_ring_kernel<CG, K, D, LNX, STATS>), dim3(a.n_items), dim3(256), lds, stream,
                       g.x, g.w, g.bias, g.s0, g.s1, g.s2, g.y, g.ln_x.stats, g.ln_x.gamma, g.ln_x.beta,
                       g.ln_s0.stats, g.ln_s0.gamma, g.ln_s0.beta, g.part, a);
    return launch_status("nbasr_grouped_conv1d_node(ring)");
}

template <int CG, int K, int D>
static int launch_ring_flavour(int variant, const GroupedArgs<float>& g, const RingDims& a, hipStream_t stream)
{
    if (g.ln_x.stats && g.part) return launch_ring<CG, K, D, true, true>(variant, g, a, stream);       // (a one-node cell; not in the search space)
    if (g.ln_x.stats) return launch_ring<CG, K, D, true, false>(variant, g, a, stream);
    if (g.part) return launch_ring<CG, K, D, false, true>(variant, g, a, stream);
    return launch_ring<CG, K, D, false, false>(variant, g, a, stream);
}

template <int CG>
static int dispatch_kd_ring(int variant, int kernel, int dilation, const GroupedArgs<float>& g, const RingDims& a, hipStream_t stream)
{
    if (kernel == 5 && dilation == 1) return launch_ring_flavour<CG, 5, 1>(variant, g, a, stream);
    if (kernel == 5 && dilation == 2) return launch_ring_flavour<CG, 5, 2>(variant, g, a, stream);
    if (kernel == 7 && dilation == 1) return launch_ring_flavour<CG, 7, 1>(variant, g, a, stream);
    if (kernel == 7 && dilation == 2) return launch_ring_flavour<CG, 7, 2>(variant, g, a, stream);
    set_error("nbasr_grouped_conv1d_node: unsupported (kernel=%d, dilation=%d); search space has k in {5,7}, d in {1,2}", kernel, dilation);
    return NBASR_EINVAL;
}

// variant: NBASR_GC_RING
int grouped_conv_f32_ring(int variant, const GroupedArgs<float>& g, int kernel, int dilation, hipStream_t stream)
{
    if (static_cast<long long>(g.ld) * 4 * 3 >= (1ll << 31)) {
        set_error("nbasr_grouped_conv1d_node: rows too long for 32-bit buffer offsets");
        return NBASR_EINVAL;
    }
    RingDims a{};
    a.channels = g.channels; a.frames = g.frames; a.ld = g.ld; a.groups = g.groups; a.batch = g.batch;
    const int nq = g.ld / 4;
    a.n_xt = (nq + 63) / 64;
    a.n_items = a.n_xt * ((g.groups + 3) / 4) * g.batch;
    a.halo = a.n_xt > 1 ? 1 : 0;
    switch (g.channels / g.groups) {
        case 6:  return dispatch_kd_ring<6>(variant, kernel, dilation, g, a, stream);
        case 8:  return dispatch_kd_ring<8>(variant, kernel, dilation, g, a, stream);
        case 10: return dispatch_kd_ring<10>(variant, kernel, dilation, g, a, stream);
        case 12: return dispatch_kd_ring<12>(variant, kernel, dilation, g, a, stream);
        default:
            set_error("nbasr_grouped_conv1d_node: channels/groups=%d unsupported; search space has 6, 8, 10, 12", g.channels / g.groups);
            return NBASR_EINVAL;
    }
}

}  // namespace nbasr
