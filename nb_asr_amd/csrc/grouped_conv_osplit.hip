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
#include "grouped_conv_impl.h"

namespace nbasr {

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

template <int CG>
static int dispatch_kd_osplit(int kernel, int dilation, const GroupedArgs<float>& a, hipStream_t stream)
{
    if (kernel == 5 && dilation == 1) return launch_osplit<CG, 5, 1>(a, stream);
    if (kernel == 5 && dilation == 2) return launch_osplit<CG, 5, 2>(a, stream);
    if (kernel == 7 && dilation == 1) return launch_osplit<CG, 7, 1>(a, stream);
    if (kernel == 7 && dilation == 2) return launch_osplit<CG, 7, 2>(a, stream);
    set_error("nbasr_grouped_conv1d_node: unsupported (kernel=%d, dilation=%d); search space has k in {5,7}, d in {1,2}", kernel, dilation);
    return NBASR_EINVAL;
}

int grouped_conv_f32_osplit(const GroupedArgs<float>& a, int kernel, int dilation, hipStream_t stream)
{
    if (a.part) {
        set_error("nbasr_grouped_conv1d_node: the output-split variant has no statistics epilogue");
        return NBASR_EINVAL;
    }
    switch (a.channels / a.groups) {
        case 6:  return dispatch_kd_osplit<6>(kernel, dilation, a, stream);
        case 8:  return dispatch_kd_osplit<8>(kernel, dilation, a, stream);
        case 10: return dispatch_kd_osplit<10>(kernel, dilation, a, stream);
        case 12: return dispatch_kd_osplit<12>(kernel, dilation, a, stream);
        default:
            set_error("nbasr_grouped_conv1d_node: channels/groups=%d unsupported; search space has 6, 8, 10, 12", a.channels / a.groups);
            return NBASR_EINVAL;
    }
}

}  // namespace nbasr
