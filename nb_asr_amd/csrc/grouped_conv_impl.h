// The fused grouped-convolution node kernel as a template over storage type / frames per lane / weight layout, and its
// dispatch over the search space's (group width, kernel, dilation).  Included by the translation units that instantiate
// one variant each (grouped_conv.hip: fp32; grouped_conv_bf16.hip: bf16; grouped_conv_alt.hip: the A/B variants), so the
// variants compile in parallel.
#pragma once
#include "storage.h"

namespace nbasr {

// LNX: the main input carries a pending LayerNorm (deferred normalisation, nbasr.h) applied while loading;
// ln_s0.stats != nullptr: skip0 carries one (inside a cell both are the cell input, with the same statistics).
// STATS: the epilogue also emits this workgroup's partial LayerNorm statistics of y -- per frame the (mean, M2) over the
// 4 x CG channels of its four groups -- to `part` ([group quad][batch][2][ld]); stats_finalize_kernel merges the quads.
// This replaces the separate statistics pass over y when y is the last node of a cell.
// T: storage type of x / skips / y (float, or bf16_t: converted to fp32 on load, rounded once on store; weights, bias,
// statistics, gamma / beta are fp32 either way).  FPL: frames per lane (4 or 8): one 16-byte access per 4 fp32 / 8 bf16
// frames.  WPERM: the weights were re-laid-out as [group][ci][tap][co] (pack_grouped_weights_kernel), so the CG * K weights
// of one input channel are one contiguous run for the scalar loads.
// FLAT: lanes are dealt over the FLATTENED (utterance, chunk) index, so a wave is full whatever the number of frames (400-frame
// bf16 rows = 50 chunks would leave 14 of 64 lanes idle); otherwise the utterance is blockIdx.z and wave-uniform (scalar
// address arithmetic: the fp32 default, whose 250-chunk rows fill 250 of 256 lanes anyway).
template <typename T, int CG, int K, int D, bool LNX, bool STATS, int FPL, bool WPERM, bool FLAT>
__global__ __launch_bounds__(256) void grouped_conv_kernel(
    const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
    const T* __restrict__ s0, const T* __restrict__ s1, const T* __restrict__ s2,
    T* __restrict__ y, int batch, int channels, int frames, int ld, int groups, const LnRef ln_x, const LnRef ln_s0,
    float* __restrict__ part)
{
    constexpr int LPAD = pad_left(K, D, 1);
    constexpr int SPAN = (K - 1) * D;                   // taps reach frames [t - LPAD, t - LPAD + SPAN]
    constexpr int QL = (LPAD + FPL - 1) / FPL;          // whole chunks left of the lane's own chunk
    constexpr int QR = (SPAN - LPAD + FPL - 1) / FPL;   // whole chunks right of it
    constexpr int NCH = QL + 1 + QR;
    constexpr int BASE = FPL * QL - LPAD;               // window index of (r = 0, tap = 0)

    const int nq = ld / FPL;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int item = blockIdx.x * 64 + lane;            // FLAT: flattened (utterance, chunk); else the chunk
    // wave-uniform group index (scalar registers => s_load for weights and bias)
    const int g = __builtin_amdgcn_readfirstlane(blockIdx.y * 4 + wave);
    if (!STATS && g >= groups) return;              // with STATS every wave must reach the workgroup barrier below
    const bool in_range = FLAT ? item < batch * nq : item < nq;
    const bool active = in_range && g < groups;
    const int b = FLAT ? (active ? item / nq : 0) : static_cast<int>(blockIdx.z);
    const int q = FLAT ? item - b * nq : item;

    // a surplus wave of the last group quad (STATS flavour, groups % 4 != 0) computes on the last group's weights and
    // stores nothing: every address below stays inside the tensors
    const int ga = g < groups ? g : groups - 1;
    const size_t row0 = (static_cast<size_t>(b) * channels + static_cast<size_t>(ga) * CG) * ld;
    const float* __restrict__ wg = w + static_cast<size_t>(ga) * (CG * CG * K);
    const float* __restrict__ bg = bias + ga * CG;

    float acc[CG][FPL];
#pragma unroll
    for (int co = 0; co < CG; ++co) {
        const float bv = bg[co];
#pragma unroll
        for (int r = 0; r < FPL; ++r) acc[co][r] = bv;
    }

    // per-frame LayerNorm statistics of the window (shared by all input channels), kept as frame PAIRS so that the
    // normalisation below is packed arithmetic: -mean, rstd and a 0/1 mask (rstd == 0 marks frames outside the utterance,
    // which must stay exactly 0).  (x + -mean) * rstd, fma(., gamma, beta), * mask rounds exactly like ln_apply.
    typedef float f2 __attribute__((ext_vector_type(2)));
    constexpr int NP = LNX ? NCH * FPL / 2 : 1;
    f2 nmw[NP], rw[NP], kw[NP];
    if (LNX) {
        const float* __restrict__ mrow = ln_x.stats + static_cast<size_t>(b) * 2 * ld;
        const float* __restrict__ rrow = mrow + ld;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int qq = q - QL + c;
            float m[FPL], r[FPL];
#pragma unroll
            for (int e = 0; e < FPL; ++e) { m[e] = 0.f; r[e] = 0.f; }
            if (active && qq >= 0 && qq < nq) { load_frames<FPL>(mrow + qq * FPL, m); load_frames<FPL>(rrow + qq * FPL, r); }
#pragma unroll
            for (int e = 0; e < FPL / 2; ++e) {
                const int pi = (c * FPL / 2 + e) % NP;
                nmw[pi] = f2{-m[2 * e], -m[2 * e + 1]};
                rw[pi] = f2{r[2 * e], r[2 * e + 1]};
                kw[pi] = f2{r[2 * e] != 0.f ? 1.f : 0.f, r[2 * e + 1] != 0.f ? 1.f : 0.f};
            }
        }
    }

#pragma unroll 1
    for (int ci = 0; ci < CG; ++ci) {
        const T* __restrict__ xrow = x + row0 + static_cast<size_t>(ci) * ld;
        float xw[NCH * FPL];
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int qq = q - QL + c;
            float v[FPL];
#pragma unroll
            for (int e = 0; e < FPL; ++e) v[e] = 0.f;
            if (active && qq >= 0 && qq < nq) load_frames<FPL>(xrow + qq * FPL, v);
#pragma unroll
            for (int e = 0; e < FPL; ++e) xw[FPL * c + e] = v[e];
        }
        if (LNX) {
            const float gam = ln_x.gamma[ga * CG + ci], bet = ln_x.beta[ga * CG + ci];     // wave-uniform: scalar loads
            const f2 gam2 = f2{gam, gam}, bet2 = f2{bet, bet};
#pragma unroll
            for (int p = 0; p < NCH * FPL / 2; ++p) {
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
                const float wv = WPERM ? wg[(ci * K + j) * CG + co] : wg[(co * CG + ci) * K + j];
#pragma unroll
                for (int r = 0; r < FPL; ++r)
                    acc[co][r] = __builtin_fmaf(wv, xw[BASE + r + j * D], acc[co][r]);
            }
        }
    }

    if (!STATS && !active) return;
    const int t0 = q * FPL;
    const bool ragged = __any(active && t0 + FPL - 1 >= frames) != 0;   // wave-uniform
    float sm[FPL], sr[FPL];                                            // statistics of this lane's own frames (skip0)
#pragma unroll
    for (int e = 0; e < FPL; ++e) { sm[e] = 0.f; sr[e] = 0.f; }
    float sk[FPL];                                                     // 0 / 1: frames beyond the utterance (rstd == 0) stay exactly 0
#pragma unroll
    for (int e = 0; e < FPL; ++e) sk[e] = 0.f;
    if (active && s0 && ln_s0.stats) {
        const float* __restrict__ mrow = ln_s0.stats + static_cast<size_t>(b) * 2 * ld;
        load_frames<FPL>(mrow + t0, sm);
        load_frames<FPL>(mrow + ld + t0, sr);
        // the mask is computed ONCE here: written as `rstd != 0 ? ... : 0` per element (ln_apply) hipcc re-evaluates the compare
        // and the subtraction for every channel of the epilogue (see grouped_conv.hip on the fp32 default)
#pragma unroll
        for (int e = 0; e < FPL; ++e) { sk[e] = sr[e] != 0.f ? 1.f : 0.f; sm[e] = -sm[e]; }
    }
    if (active) {
#pragma unroll
    for (int co = 0; co < CG; ++co) {
        const size_t off = row0 + static_cast<size_t>(co) * ld + t0;
        float o[FPL];
#pragma unroll
        for (int r = 0; r < FPL; ++r) o[r] = relu_clamp(acc[co][r]);
        if (s0) {
            float v[FPL];
            load_frames<FPL>(s0 + off, v);
            if (ln_s0.stats) {
                const float gam = ln_s0.gamma[ga * CG + co], bet = ln_s0.beta[ga * CG + co];
#pragma unroll
                for (int r = 0; r < FPL; ++r) v[r] = __builtin_fmaf((v[r] + sm[r]) * sr[r], gam, bet) * sk[r];    // == ln_apply for finite values
            }
#pragma unroll
            for (int r = 0; r < FPL; ++r) o[r] += v[r];
        }
        if (s1) {
            float v[FPL];
            load_frames<FPL>(s1 + off, v);
#pragma unroll
            for (int r = 0; r < FPL; ++r) o[r] += v[r];
        }
        if (s2) {
            float v[FPL];
            load_frames<FPL>(s2 + off, v);
#pragma unroll
            for (int r = 0; r < FPL; ++r) o[r] += v[r];
        }
        // keep the pitch columns frames..ld-1 at zero (layout invariant, nbasr.h); only the wave that holds the ragged chunk
        if (ragged) {
#pragma unroll
            for (int r = 0; r < FPL; ++r) if (t0 + r >= frames) o[r] = 0.f;
        }
        // streaming (non-temporal) stores: worth 2 % of the forward over plain ones (round 1).  Measured and rejected in round 2: a
        // RUN-TIME choice between the two store flavours (plain for tensors that fit the 256 MiB last-level cache) -- no gain
        // from the plain stores, and the second store path alone made this kernel 30 % slower in the widest-row block
        store_frames<FPL, true>(y + off, o);
        if (STATS) {
#pragma unroll
            for (int r = 0; r < FPL; ++r) acc[co][r] = o[r];         // keep the final values for the statistics
        }
    }
    }
    if (STATS) {
        // per-lane (mean, M2) over this group's CG channels, exact two-pass in registers (of the fp32 values: with bf16
        // storage the statistics describe the tensor before its rounding, to within bf16's 2^-9 relative)
        __shared__ float sp[4][2 * FPL][64];
        float pm[FPL], p2[FPL];
#pragma unroll
        for (int r = 0; r < FPL; ++r) {
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
        for (int r = 0; r < FPL; ++r) { sp[wave][r][lane] = pm[r]; sp[wave][FPL + r][lane] = p2[r]; }
        __syncthreads();
        if (wave == 0 && in_range) {
            const int g0 = blockIdx.y * 4;
            const int nw = min(4, groups - g0);                      // groups (waves) that hold real data
            float om[FPL], o2[FPL];
#pragma unroll
            for (int r = 0; r < FPL; ++r) {
                float mean = 0.f;
                for (int k = 0; k < nw; ++k) mean += sp[k][r][lane];
                mean /= static_cast<float>(nw);
                float m2 = 0.f;
                for (int k = 0; k < nw; ++k) { const float d = sp[k][r][lane] - mean; m2 += sp[k][FPL + r][lane] + CG * d * d; }
                om[r] = mean; o2[r] = m2;
            }
            float* prow = part + (static_cast<size_t>(blockIdx.y) * batch + b) * 2 * ld + t0;
            store_frames<FPL, false>(prow, om);
            store_frames<FPL, false>(prow + ld, o2);
        }
    }
}


template <typename T>
struct GroupedArgs {
    const T* x; const float* w; const float* bias; const T* s0; const T* s1; const T* s2; T* y;
    int batch, channels, frames, ld, groups;
    LnRef ln_x, ln_s0;
    float* part;                 // partial-statistics workspace (nullptr: no statistics)
};

#ifndef NBASR_GC_FLAT_F32
#define NBASR_GC_FLAT_F32 0      /* fp32 kernels: utterance per blockIdx.z (1) or flattened lanes (0/1 A/B at build time) */
#endif
template <typename T> constexpr bool gc_flat() { return sizeof(T) == 2 || NBASR_GC_FLAT_F32 != 0; }

template <typename T, int FPL, bool WPERM, int CG, int K, int D>
static int launch_grouped(const GroupedArgs<T>& a, hipStream_t stream)
{
    constexpr bool FLAT = gc_flat<T>();
    const long long items = static_cast<long long>(FLAT ? a.batch : 1) * (a.ld / FPL);
    dim3 grid(static_cast<unsigned>((items + 63) / 64), (a.groups + 3) / 4, FLAT ? 1 : a.batch);
#define NBASR_LAUNCH_GROUPED(LNX, STATS)                                                                                    \
    hipLaunchKernelGGL((grouped_conv_kernel<T, CG, K, D, LNX, STATS, FPL, WPERM, FLAT>), grid, dim3(256), 0, stream, a.x, a.w, a.bias, \
                       a.s0, a.s1, a.s2, a.y, a.batch, a.channels, a.frames, a.ld, a.groups, a.ln_x, a.ln_s0, a.part)
    if (a.ln_x.stats) { if (a.part) NBASR_LAUNCH_GROUPED(true, true); else NBASR_LAUNCH_GROUPED(true, false); }
    else              { if (a.part) NBASR_LAUNCH_GROUPED(false, true); else NBASR_LAUNCH_GROUPED(false, false); }
#undef NBASR_LAUNCH_GROUPED
    return launch_status("nbasr_grouped_conv1d_node");
}

template <typename T, int FPL, bool WPERM, int CG>
static int dispatch_kd(int kernel, int dilation, const GroupedArgs<T>& a, hipStream_t stream)
{
    if (kernel == 5 && dilation == 1) return launch_grouped<T, FPL, WPERM, CG, 5, 1>(a, stream);
    if (kernel == 5 && dilation == 2) return launch_grouped<T, FPL, WPERM, CG, 5, 2>(a, stream);
    if (kernel == 7 && dilation == 1) return launch_grouped<T, FPL, WPERM, CG, 7, 1>(a, stream);
    if (kernel == 7 && dilation == 2) return launch_grouped<T, FPL, WPERM, CG, 7, 2>(a, stream);
    set_error("nbasr_grouped_conv1d_node: unsupported (kernel=%d, dilation=%d); search space has k in {5,7}, d in {1,2}", kernel, dilation);
    return NBASR_EINVAL;
}

// one variant (storage type, frames per lane, weight layout) over every group width of the model
template <typename T, int FPL, bool WPERM>
int grouped_conv_variant(const GroupedArgs<T>& a, int kernel, int dilation, hipStream_t stream)
{
    switch (a.channels / a.groups) {
        case 6:  return dispatch_kd<T, FPL, WPERM, 6>(kernel, dilation, a, stream);
        case 8:  return dispatch_kd<T, FPL, WPERM, 8>(kernel, dilation, a, stream);
        case 10: return dispatch_kd<T, FPL, WPERM, 10>(kernel, dilation, a, stream);
        case 12: return dispatch_kd<T, FPL, WPERM, 12>(kernel, dilation, a, stream);
        default:
            set_error("nbasr_grouped_conv1d_node: channels/groups=%d unsupported (model widths give 6, 8, 10, 12)", a.channels / a.groups);
            return NBASR_EINVAL;
    }
}

// defined one per translation unit (explicit variants)
int grouped_conv_f32_base(const GroupedArgs<float>& a, int kernel, int dilation, hipStream_t stream);     // FPL 4, torch weight layout
int grouped_conv_f32_alt(int variant, const GroupedArgs<float>& a, int kernel, int dilation, hipStream_t stream);
int grouped_conv_f32_osplit(int variant, const GroupedArgs<float>& a, int kernel, int dilation, hipStream_t stream);      // grouped_conv_osplit.hip
int grouped_conv_f32_ring(int variant, const GroupedArgs<float>& a, int kernel, int dilation, hipStream_t stream);        // grouped_conv_ring.hip
int grouped_conv_bf16(int variant, const GroupedArgs<bf16_t>& a, int kernel, int dilation, hipStream_t stream);

}  // namespace nbasr
