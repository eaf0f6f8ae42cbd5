// Output-split form of the fp32 node kernel (variant bit NBASR_GC_OSPLIT): a wave owns (utterance, group, HALF of the group's output
// channels) instead of the whole group.
//
// Why: the default kernel (grouped_conv.hip) gives a wave 256 frames x all CG output channels of a group.  With short rows there
// are few such waves -- C = 1200 at 250 frames and 64 utterances is 6 400 waves = 6.25 per SIMD at 6 resident (73 registers), C = 1200
// at 8 utterances is 800 waves on 1 024 SIMDs -- so the launch is one wave's dependent chain of CG window loads, not a stream.
// Halving the accumulators (24 instead of 48 at CG = 12) makes twice as many waves of half the work at 10-11 resident per SIMD;
// each input window is loaded by both halves (the second one hits in L1/L2, HBM still sees every byte once).  Every output is the
// same sum in the same order: bit-identical to the default kernel.  No statistics flavour (the last node of a cell keeps the
// default kernel, whose epilogue partials are per whole group).
//
// PIPE (variant bit NBASR_GC_PIPE, with or without the output split): software-pipelined window loads.  In the default kernel a
// wave's channel loop is  {load window -> s_waitcnt vmcnt(0) -> scalar weight loads -> wait -> 120 packed FMAs}: nothing of a wave's
// own overlaps, only other waves hide its latency (6-8 per SIMD).  Here the window of channel ci + 1 is requested BEFORE the FMAs of
// channel ci, two register sets in ping-pong over a channel loop unrolled by two.  That only works branch-free (hipcc places a full
// vmcnt(0) at every control-flow join, which is what defeated the earlier prefetch experiment), so the window comes through
// BUFFER loads: one descriptor per input row (base = the row, num_records = its pitch), out-of-range chunks -- left of frame 0,
// right of the row, lanes beyond the row -- return zeros from the hardware's bounds check: no predicate, no branch, no mask.
#include "grouped_conv_impl.h"

namespace nbasr {

typedef float gc_f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float4 row_chunk(const float* row, int row_bytes, int byte_offset)
{
    // raw buffer resource over ONE row: word 3 = 0x00020000 (gfx9 raw buffer, 32-bit data format unused), bounds-checked.
    // (Cast the WHOLE result: taking .x/.y/.z/.w of the builtin's integer vector made hipcc 7.2 load a single dword.)
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(row), 0, row_bytes, 0x00020000);
    const gc_f4 f = __builtin_bit_cast(gc_f4, __builtin_amdgcn_raw_buffer_load_b128(rs, byte_offset, 0, 0));
    return make_float4(f[0], f[1], f[2], f[3]);
}

// STATS (OS == 1 only): the epilogue also emits the workgroup's partial LayerNorm statistics of y, exactly as the default kernel
// does (`part`: [group quad][batch][2][ld]; merged by stats_finalize_kernel).
template <int CG, int K, int D, bool LNX, int OS, bool STATS>
__global__ __launch_bounds__(256) void grouped_conv_f32_pipe_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
    const float* __restrict__ s0, const float* __restrict__ s1, const float* __restrict__ s2,
    float* __restrict__ y, int channels, int frames, int ld, int groups, const LnRef ln_x, const LnRef ln_s0,
    float* __restrict__ part)
{
    static_assert(CG % 2 == 0 && (OS == 1 || OS == 2), "channel loop unrolled by two; one or two waves per group");
    static_assert(!STATS || OS == 1, "the statistics partials are per whole group");
    constexpr int CO = CG / OS;                  // output channels of this wave
    constexpr int LPAD = pad_left(K, D, 1);
    constexpr int SPAN = (K - 1) * D;
    constexpr int QL = (LPAD + 3) / 4;
    constexpr int QR = (SPAN - LPAD + 3) / 4;
    constexpr int NCH = QL + 1 + QR;
    constexpr int BASE = 4 * QL - LPAD;
    constexpr int GPW = 4 / OS;                  // groups per 4-wave workgroup

    const int nq = ld >> 2;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = blockIdx.x * 64 + lane;
    const int g_raw = __builtin_amdgcn_readfirstlane(blockIdx.y * GPW + wave / OS);
    const int co0 = (wave % OS) * CO;
    const int b = blockIdx.z;
    if (!STATS && g_raw >= groups) return;       // wave-uniform; with STATS every wave must reach the workgroup barrier below:
    const int g = g_raw < groups ? g_raw : groups - 1;      // a surplus wave recomputes the last group and stores nothing

    const size_t row0 = (static_cast<size_t>(b) * channels + static_cast<size_t>(g) * CG) * ld;
    const float* __restrict__ wg = w + (static_cast<size_t>(g) * CG + co0) * (CG * K);
    const float* __restrict__ bg = bias + g * CG + co0;
    const int row_bytes = ld * 4;
    const int off0 = (q - QL) * 16;              // byte offset of the window's first chunk in a row (negative / beyond: zeros)

    float acc[CO][4];
#pragma unroll
    for (int co = 0; co < CO; ++co) {
        const float bv = bg[co];
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[co][r] = bv;
    }

    typedef float f2 __attribute__((ext_vector_type(2)));
    constexpr int NP = LNX ? NCH * 2 : 1;
    f2 nmw[NP], rw[NP], kw[NP];
    if (LNX) {
        const float* __restrict__ mrow = ln_x.stats + static_cast<size_t>(b) * 2 * ld;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const float4 m = row_chunk(mrow, row_bytes, off0 + 16 * c), r = row_chunk(mrow + ld, row_bytes, off0 + 16 * c);
            nmw[(2 * c) % NP] = f2{-m.x, -m.y}; nmw[(2 * c + 1) % NP] = f2{-m.z, -m.w};
            rw[(2 * c) % NP] = f2{r.x, r.y};    rw[(2 * c + 1) % NP] = f2{r.z, r.w};
            kw[(2 * c) % NP] = f2{r.x != 0.f ? 1.f : 0.f, r.y != 0.f ? 1.f : 0.f};
            kw[(2 * c + 1) % NP] = f2{r.z != 0.f ? 1.f : 0.f, r.w != 0.f ? 1.f : 0.f};
        }
    }

    const float* __restrict__ xg = x + row0;
    auto fetch = [&](int ci, float4 (&dst)[NCH]) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) dst[c] = row_chunk(xg + static_cast<size_t>(ci) * ld, row_bytes, off0 + 16 * c);
    };
    auto consume = [&](int ci, const float4 (&src)[NCH]) {
        float xw[NCH * 4];
#pragma unroll
        for (int c = 0; c < NCH; ++c) { xw[4 * c + 0] = src[c].x; xw[4 * c + 1] = src[c].y; xw[4 * c + 2] = src[c].z; xw[4 * c + 3] = src[c].w; }
        if (LNX) {
            const float gam = ln_x.gamma[g * CG + ci], bet = ln_x.beta[g * CG + ci];
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
            for (int co = 0; co < CO; ++co) {
                const float wv = wg[(co * CG + ci) * K + j];
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    acc[co][r] = __builtin_fmaf(wv, xw[BASE + r + j * D], acc[co][r]);
            }
        }
    };

    float4 wa[NCH], wb[NCH];
    fetch(0, wa);
#pragma unroll 1
    for (int ci = 0; ci < CG; ci += 2) {
        fetch(ci + 1, wb);
        consume(ci, wa);
        fetch(ci + 2 < CG ? ci + 2 : CG - 1, wa);           // last round: a redundant reload instead of a branch
        consume(ci + 1, wb);
    }

    if (!STATS && q >= nq) return;
    const bool active = q < nq;
    const int t0 = q * 4;
    const bool ragged = __any(active && t0 + 3 >= frames) != 0;   // wave-uniform
    float4 sm = make_float4(0.f, 0.f, 0.f, 0.f), sr = sm;          // statistics of this lane's own 4 frames (skip0)
    if (active && s0 && ln_s0.stats) {
        const float4* __restrict__ mrow = reinterpret_cast<const float4*>(ln_s0.stats + static_cast<size_t>(b) * 2 * ld);
        sm = mrow[q];
        sr = mrow[nq + q];
    }
    if (active) {
#pragma unroll
    for (int co = 0; co < CO; ++co) {
        const size_t off = row0 + static_cast<size_t>(co0 + co) * ld + t0;
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = relu_clamp(acc[co][r]);
        if (s0) {
            float4 v = *reinterpret_cast<const float4*>(s0 + off);
            if (ln_s0.stats) {
                const float gam = ln_s0.gamma[g * CG + co0 + co], bet = ln_s0.beta[g * CG + co0 + co];
                v.x = ln_apply(v.x, sm.x, sr.x, gam, bet); v.y = ln_apply(v.y, sm.y, sr.y, gam, bet);
                v.z = ln_apply(v.z, sm.z, sr.z, gam, bet); v.w = ln_apply(v.w, sm.w, sr.w, gam, bet);
            }
            o[0] += v.x; o[1] += v.y; o[2] += v.z; o[3] += v.w;
        }
        if (s1) { const float4 v = *reinterpret_cast<const float4*>(s1 + off); o[0] += v.x; o[1] += v.y; o[2] += v.z; o[3] += v.w; }
        if (s2) { const float4 v = *reinterpret_cast<const float4*>(s2 + off); o[0] += v.x; o[1] += v.y; o[2] += v.z; o[3] += v.w; }
        if (ragged) {
#pragma unroll
            for (int r = 0; r < 4; ++r) if (t0 + r >= frames) o[r] = 0.f;
        }
        typedef float f4v __attribute__((ext_vector_type(4)));
        if (!STATS || g_raw < groups)
            __builtin_nontemporal_store(f4v{o[0], o[1], o[2], o[3]}, reinterpret_cast<f4v*>(y + off));
        if (STATS) {
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[co][r] = o[r];           // keep the final values for the statistics
        }
    }
    }
    if constexpr (STATS) {
        // per-lane (mean, M2) over this group's CG channels, exact two-pass in registers; wave 0 merges the workgroup's groups
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
        if (wave == 0 && q < nq) {
            const int g0 = blockIdx.y * 4;
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
            float* prow = part + (static_cast<size_t>(blockIdx.y) * gridDim.z + b) * 2 * ld + t0;
            *reinterpret_cast<float4*>(prow) = make_float4(om[0], om[1], om[2], om[3]);
            *reinterpret_cast<float4*>(prow + ld) = make_float4(o2[0], o2[1], o2[2], o2[3]);
        }
    }
}

template <int CG, int K, int D, bool LNX>
__global__ __launch_bounds__(256) void grouped_conv_f32_osplit_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
    const float* __restrict__ s0, const float* __restrict__ s1, const float* __restrict__ s2,
    float* __restrict__ y, int channels, int frames, int ld, int groups, const LnRef ln_x, const LnRef ln_s0)
{
    static_assert(CG % 2 == 0, "two halves");
    constexpr int CO = CG / 2;                   // output channels of this wave
    constexpr int LPAD = pad_left(K, D, 1);
    constexpr int SPAN = (K - 1) * D;
    constexpr int QL = (LPAD + 3) / 4;
    constexpr int QR = (SPAN - LPAD + 3) / 4;
    constexpr int NCH = QL + 1 + QR;
    constexpr int BASE = 4 * QL - LPAD;

    const int nq = ld >> 2;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = blockIdx.x * 64 + lane;
    const int g = __builtin_amdgcn_readfirstlane(blockIdx.y * 2 + (wave >> 1));       // two groups per workgroup, two waves per group
    const int co0 = (wave & 1) * CO;
    const int b = blockIdx.z;
    if (g >= groups || q >= nq) return;

    const size_t row0 = (static_cast<size_t>(b) * channels + static_cast<size_t>(g) * CG) * ld;
    const float* __restrict__ wg = w + (static_cast<size_t>(g) * CG + co0) * (CG * K);
    const float* __restrict__ bg = bias + g * CG + co0;

    float acc[CO][4];
#pragma unroll
    for (int co = 0; co < CO; ++co) {
        const float bv = bg[co];
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[co][r] = bv;
    }

    typedef float f2 __attribute__((ext_vector_type(2)));
    constexpr int NP = LNX ? NCH * 2 : 1;
    f2 nmw[NP], rw[NP], kw[NP];
    if (LNX) {
        const float4* __restrict__ mrow = reinterpret_cast<const float4*>(ln_x.stats + static_cast<size_t>(b) * 2 * ld);
        const float4* __restrict__ rrow = mrow + nq;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int qq = q - QL + c;
            float4 m = make_float4(0.f, 0.f, 0.f, 0.f), r = m;
            if (qq >= 0 && qq < nq) { m = mrow[qq]; r = rrow[qq]; }
            nmw[(2 * c) % NP] = f2{-m.x, -m.y}; nmw[(2 * c + 1) % NP] = f2{-m.z, -m.w};
            rw[(2 * c) % NP] = f2{r.x, r.y};    rw[(2 * c + 1) % NP] = f2{r.z, r.w};
            kw[(2 * c) % NP] = f2{r.x != 0.f ? 1.f : 0.f, r.y != 0.f ? 1.f : 0.f};
            kw[(2 * c + 1) % NP] = f2{r.z != 0.f ? 1.f : 0.f, r.w != 0.f ? 1.f : 0.f};
        }
    }

#pragma unroll 1
    for (int ci = 0; ci < CG; ++ci) {
        const float4* __restrict__ xrow = reinterpret_cast<const float4*>(x + row0 + static_cast<size_t>(ci) * ld);
        float xw[NCH * 4];
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int qq = q - QL + c;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (qq >= 0 && qq < nq) v = xrow[qq];
            xw[4 * c + 0] = v.x; xw[4 * c + 1] = v.y; xw[4 * c + 2] = v.z; xw[4 * c + 3] = v.w;
        }
        if (LNX) {
            const float gam = ln_x.gamma[g * CG + ci], bet = ln_x.beta[g * CG + ci];
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
            for (int co = 0; co < CO; ++co) {
                const float wv = wg[(co * CG + ci) * K + j];
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    acc[co][r] = __builtin_fmaf(wv, xw[BASE + r + j * D], acc[co][r]);
            }
        }
    }

    const int t0 = q * 4;
    const bool ragged = __any(t0 + 3 >= frames) != 0;              // wave-uniform
    float4 sm = make_float4(0.f, 0.f, 0.f, 0.f), sr = sm;          // statistics of this lane's own 4 frames (skip0)
    if (s0 && ln_s0.stats) {
        const float4* __restrict__ mrow = reinterpret_cast<const float4*>(ln_s0.stats + static_cast<size_t>(b) * 2 * ld);
        sm = mrow[q];
        sr = mrow[nq + q];
    }
#pragma unroll
    for (int co = 0; co < CO; ++co) {
        const size_t off = row0 + static_cast<size_t>(co0 + co) * ld + t0;
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = relu_clamp(acc[co][r]);
        if (s0) {
            float4 v = *reinterpret_cast<const float4*>(s0 + off);
            if (ln_s0.stats) {
                const float gam = ln_s0.gamma[g * CG + co0 + co], bet = ln_s0.beta[g * CG + co0 + co];
                v.x = ln_apply(v.x, sm.x, sr.x, gam, bet); v.y = ln_apply(v.y, sm.y, sr.y, gam, bet);
                v.z = ln_apply(v.z, sm.z, sr.z, gam, bet); v.w = ln_apply(v.w, sm.w, sr.w, gam, bet);
            }
            o[0] += v.x; o[1] += v.y; o[2] += v.z; o[3] += v.w;
        }
        if (s1) { const float4 v = *reinterpret_cast<const float4*>(s1 + off); o[0] += v.x; o[1] += v.y; o[2] += v.z; o[3] += v.w; }
        if (s2) { const float4 v = *reinterpret_cast<const float4*>(s2 + off); o[0] += v.x; o[1] += v.y; o[2] += v.z; o[3] += v.w; }
        if (ragged) {
#pragma unroll
            for (int r = 0; r < 4; ++r) if (t0 + r >= frames) o[r] = 0.f;
        }
        typedef float f4v __attribute__((ext_vector_type(4)));
        __builtin_nontemporal_store(f4v{o[0], o[1], o[2], o[3]}, reinterpret_cast<f4v*>(y + off));
    }
}

template <int CG, int K, int D>
static int launch_osplit(const GroupedArgs<float>& a, hipStream_t stream)
{
    const int nq = a.ld / 4;
    const dim3 grid((nq + 63) / 64, (a.groups + 1) / 2, a.batch);
    if (a.ln_x.stats)
        hipLaunchKernelGGL((grouped_conv_f32_osplit_kernel<CG, K, D, true>), grid, dim3(256), 0, stream, a.x, a.w, a.bias, a.s0, a.s1, a.s2,
                           a.y, a.channels, a.frames, a.ld, a.groups, a.ln_x, a.ln_s0);
    else
        hipLaunchKernelGGL((grouped_conv_f32_osplit_kernel<CG, K, D, false>), grid, dim3(256), 0, stream, a.x, a.w, a.bias, a.s0, a.s1, a.s2,
                           a.y, a.channels, a.frames, a.ld, a.groups, a.ln_x, a.ln_s0);
    return launch_status("nbasr_grouped_conv1d_node(osplit)");
}

template <int CG, int K, int D, int OS, bool STATS>
static int launch_pipe(const GroupedArgs<float>& a, hipStream_t stream)
{
    const int nq = a.ld / 4;
    constexpr int GPW = 4 / OS;
    const dim3 grid((nq + 63) / 64, (a.groups + GPW - 1) / GPW, a.batch);
    if (a.ln_x.stats)
        hipLaunchKernelGGL((grouped_conv_f32_pipe_kernel<CG, K, D, true, OS, STATS>), grid, dim3(256), 0, stream, a.x, a.w, a.bias, a.s0, a.s1, a.s2,
                           a.y, a.channels, a.frames, a.ld, a.groups, a.ln_x, a.ln_s0, a.part);
    else
        hipLaunchKernelGGL((grouped_conv_f32_pipe_kernel<CG, K, D, false, OS, STATS>), grid, dim3(256), 0, stream, a.x, a.w, a.bias, a.s0, a.s1, a.s2,
                           a.y, a.channels, a.frames, a.ld, a.groups, a.ln_x, a.ln_s0, a.part);
    return launch_status("nbasr_grouped_conv1d_node(pipe)");
}

template <int CG, int K, int D>
static int launch_alt2(int variant, const GroupedArgs<float>& a, hipStream_t stream)
{
    if (!(variant & NBASR_GC_PIPE)) return launch_osplit<CG, K, D>(a, stream);
    if (variant & NBASR_GC_OSPLIT) return launch_pipe<CG, K, D, 2, false>(a, stream);
    return a.part ? launch_pipe<CG, K, D, 1, true>(a, stream) : launch_pipe<CG, K, D, 1, false>(a, stream);
}

template <int CG>
static int dispatch_kd_alt2(int variant, int kernel, int dilation, const GroupedArgs<float>& a, hipStream_t stream)
{
    if (kernel == 5 && dilation == 1) return launch_alt2<CG, 5, 1>(variant, a, stream);
    if (kernel == 5 && dilation == 2) return launch_alt2<CG, 5, 2>(variant, a, stream);
    if (kernel == 7 && dilation == 1) return launch_alt2<CG, 7, 1>(variant, a, stream);
    if (kernel == 7 && dilation == 2) return launch_alt2<CG, 7, 2>(variant, a, stream);
    set_error("nbasr_grouped_conv1d_node: unsupported (kernel=%d, dilation=%d); search space has k in {5,7}, d in {1,2}", kernel, dilation);
    return NBASR_EINVAL;
}

// variant: NBASR_GC_OSPLIT, NBASR_GC_PIPE or both
int grouped_conv_f32_osplit(int variant, const GroupedArgs<float>& a, int kernel, int dilation, hipStream_t stream)
{
    if (a.part && (variant & NBASR_GC_OSPLIT)) {
        set_error("nbasr_grouped_conv1d_node: the output-split variants have no statistics epilogue");
        return NBASR_EINVAL;
    }
    if ((variant & NBASR_GC_PIPE) && static_cast<long long>(a.ld) * 4 * 3 >= (1ll << 31)) {
        set_error("nbasr_grouped_conv1d_node: rows too long for 32-bit buffer offsets");
        return NBASR_EINVAL;
    }
    switch (a.channels / a.groups) {
        case 6:  return dispatch_kd_alt2<6>(variant, kernel, dilation, a, stream);
        case 8:  return dispatch_kd_alt2<8>(variant, kernel, dilation, a, stream);
        case 10: return dispatch_kd_alt2<10>(variant, kernel, dilation, a, stream);
        case 12: return dispatch_kd_alt2<12>(variant, kernel, dilation, a, stream);
        default:
            set_error("nbasr_grouped_conv1d_node: channels/groups=%d unsupported; search space has 6, 8, 10, 12", a.channels / a.groups);
            return NBASR_EINVAL;
    }
}

}  // namespace nbasr
