// Fused node operation for the grouped convolutions of the search space:
//     y = min(relu(conv1d(zero_pad(x), w, bias, dilation, groups)), 20) + skip0 + skip1 + skip2
// replacing reference ops.py:24-30 (ZeroPad2d -> Conv1d -> ReLU -> clamp_max_) and the python `sum`
// of Node.forward (model.py:13-22).  No padded copy, no zeros_like tensors, one pass over HBM.
//
// Mapping (gfx950, wave64): one lane owns 4 consecutive frames (one 16-byte chunk) of ALL channels
// of one channel group; a wave covers 256 consecutive frames of one (utterance, group), so every
// global access is a fully coalesced 1 KiB wave transaction, and the group's weights / bias are
// wave-uniform and come through the scalar cache (s_load), never through VGPRs or LDS.
// The k-tap sliding window of an input channel lives in registers: NCH aligned 16-byte chunks per
// lane (neighbour lanes re-read the halo chunks from L1/L2, HBM sees every byte once).
//
// HBM-bound (7.5-21 flop/byte); algorithmic bytes per launch
//     4 * (B*C*T * (2 + n_skips) + C*(C/groups)*k + C).
#include "grouped_conv_impl.h"

namespace nbasr {

// weights (channels, CG, K) -> [group][ci][tap][co]: the CG * K weights of one input channel become ONE contiguous run
// (wide scalar loads instead of one per output channel); done once per weight version by the caller
__global__ __launch_bounds__(256) void pack_grouped_weights_kernel(const float* __restrict__ w, float* __restrict__ wp,
                                                                   int channels, int cg, int k)
{
    const int n = channels * cg * k;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        int e = i;
        const int co = e % cg; e /= cg;
        const int j = e % k; e /= k;
        const int ci = e % cg; e /= cg;
        const int g = e;
        wp[i] = w[((static_cast<size_t>(g) * cg + co) * cg + ci) * k + j];
    }
}

// merge the partial statistics: stats (batch, 2, ld) <- (mean, rstd) per frame, 0 in the pitch columns.
// A workgroup = 32 frames x 8 part lanes: lane p merges the partials p, p + 8, ... (all its loads issued together: one round trip),
// then one thread per frame merges the 8 lane results from LDS.  The first version walked all 25-50 partials of a frame in ONE
// thread, five at a time: 5-10 dependent round trips = 10-12 us per launch whatever the batch, 15 launches per forward
// (0.18 ms: 10 % of a step at 8 utterances per GPU).  Chan's merge throughout; the order is fixed, so results are reproducible.
// Round 5: a thread takes FOUR consecutive frames (16-byte loads: a wave reads 512 contiguous bytes per partial row instead of 128) and
// only the partials that exist (the clamped duplicates of the first version were 19 % of the loads at 100 partials, 75 % at 25); per
// frame the merges and their order are the first version's, so the statistics are the same bit for bit.  51 MB of per-group partials:
// 20 -> see profiles/NOTES_r05.md.
constexpr int SF_QUADS = 32, SF_FRAMES = 4 * SF_QUADS, SF_LANES = 8, SF_MAX_PER_LANE = 16;      // <= 128 partials per frame (per-group partials of a fused cell: 100)
__global__ __launch_bounds__(SF_QUADS * SF_LANES) void stats_finalize_kernel(const float* __restrict__ part, float* __restrict__ stats,
                                                                               int batch, int frames, int ld, int groups, int cg, int gpp, float eps)
{
    __shared__ float s_cnt[SF_LANES][SF_FRAMES], s_mean[SF_LANES][SF_FRAMES], s_m2[SF_LANES][SF_FRAMES];
    const int tq = threadIdx.x & (SF_QUADS - 1), pl = threadIdx.x / SF_QUADS;        // frame quads fastest
    const int t0 = blockIdx.x * SF_FRAMES + 4 * tq;
    const int b = blockIdx.y;
    const int nparts = (groups + gpp - 1) / gpp;          // partials per frame: one per `gpp` groups (4: node kernels; 2 / 1: fused cells; 16 channels: dense convs)
    float cnt[4] = {0.f, 0.f, 0.f, 0.f}, mean[4] = {0.f, 0.f, 0.f, 0.f}, m2[4] = {0.f, 0.f, 0.f, 0.f};
    if (t0 < frames) {                                    // (ld % 4 == 0: the quad lies inside the row; its frames beyond `frames` are discarded below)
        float4 pm[SF_MAX_PER_LANE], pq[SF_MAX_PER_LANE];
#pragma unroll
        for (int u = 0; u < SF_MAX_PER_LANE; ++u) {
            const int k = pl + u * SF_LANES;
            pm[u] = make_float4(0.f, 0.f, 0.f, 0.f); pq[u] = pm[u];
            if (k < nparts) {
                const float* prow = part + (static_cast<size_t>(k) * batch + b) * 2 * ld + t0;
                pm[u] = *reinterpret_cast<const float4*>(prow);
                pq[u] = *reinterpret_cast<const float4*>(prow + ld);
            }
        }
#pragma unroll
        for (int u = 0; u < SF_MAX_PER_LANE; ++u) {
            const int k = pl + u * SF_LANES;
            if (k < nparts) {
                const float nb = static_cast<float>(cg * min(gpp, groups - gpp * k));
                const float pmv[4] = {pm[u].x, pm[u].y, pm[u].z, pm[u].w}, pqv[4] = {pq[u].x, pq[u].y, pq[u].z, pq[u].w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float tot = cnt[r] + nb;
                    const float delta = pmv[r] - mean[r];
                    mean[r] += delta * (nb / tot);
                    m2[r] += pqv[r] + delta * delta * (cnt[r] * nb / tot);
                    cnt[r] = tot;
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) { s_cnt[pl][4 * tq + r] = cnt[r]; s_mean[pl][4 * tq + r] = mean[r]; s_m2[pl][4 * tq + r] = m2[r]; }
    __syncthreads();
    const int tf = threadIdx.x, t = blockIdx.x * SF_FRAMES + tf;                     // one thread per frame merges the 8 lane results, lane 0's first
    if (tf >= SF_FRAMES || t >= ld) return;
    float* srow = stats + static_cast<size_t>(b) * 2 * ld;
    if (t >= frames) { srow[t] = 0.f; srow[ld + t] = 0.f; return; }
    float c = s_cnt[0][tf], mu = s_mean[0][tf], q = s_m2[0][tf];
#pragma unroll
    for (int p = 1; p < SF_LANES; ++p) {
        const float nb = s_cnt[p][tf];
        if (nb > 0.f) {
            const float tot = c + nb;
            const float delta = s_mean[p][tf] - mu;
            mu += delta * (nb / tot);
            q += s_m2[p][tf] + delta * delta * (c * nb / tot);
            c = tot;
        }
    }
    srow[t] = mu;
    srow[ld + t] = 1.0f / sqrtf(q / c + eps);
}

// y = 0 + skip0 + skip1 + skip2 for a node whose main op is `zero` (reference ops.py:67-68); skip0 may carry a pending
// LayerNorm, in which case the materialised (normalised) value is what gets summed and stored
template <typename T>
__global__ __launch_bounds__(256) void skip_sum_kernel(
    const T* __restrict__ s0, const T* __restrict__ s1, const T* __restrict__ s2,
    T* __restrict__ y, size_t nchunks, int channels, int nq, const LnRef ln_s0)
{
    constexpr int FR = Chunk<T>::FR;
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < nchunks; i += stride) {
        float o[FR];
#pragma unroll
        for (int e = 0; e < FR; ++e) o[e] = 0.f;
        if (s0) {
            float v[FR];
            load_frames<FR>(s0 + i * FR, v);
            if (ln_s0.stats) {
                const size_t row = i / nq;
                const int q = static_cast<int>(i - row * nq);
                const int c = static_cast<int>(row % channels);
                const size_t b = row / channels;
                const float* mrow = ln_s0.stats + b * 2 * (static_cast<size_t>(nq) * FR);
                float m[FR], r[FR];
                load_frames<FR>(mrow + q * FR, m);
                load_frames<FR>(mrow + static_cast<size_t>(nq) * FR + q * FR, r);
                const float gam = ln_s0.gamma[c], bet = ln_s0.beta[c];
#pragma unroll
                for (int e = 0; e < FR; ++e) v[e] = ln_apply(v[e], m[e], r[e], gam, bet);
            }
#pragma unroll
            for (int e = 0; e < FR; ++e) o[e] += v[e];
        }
        if (s1) {
            float v[FR];
            load_frames<FR>(s1 + i * FR, v);
#pragma unroll
            for (int e = 0; e < FR; ++e) o[e] += v[e];
        }
        if (s2) {
            float v[FR];
            load_frames<FR>(s2 + i * FR, v);
#pragma unroll
            for (int e = 0; e < FR; ++e) o[e] += v[e];
        }
        store_frames<FR, false>(y + i * FR, o);
    }
}

template <typename E>          // E: any trivially copyable element (float; unsigned short for bf16 bits)
__global__ __launch_bounds__(256) void repitch_kernel(
    const E* __restrict__ src, E* __restrict__ dst, int rows, int frames, int ld_src, int ld_dst)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ld_dst) return;
    for (int r = blockIdx.y; r < rows; r += gridDim.y)
        dst[static_cast<size_t>(r) * ld_dst + t] = (t < frames) ? src[static_cast<size_t>(r) * ld_src + t] : E(0);
}

// ---- the fp32 default: the round-1 kernel, kept verbatim ---------------------------------------------------------------------
// grouped_conv_impl.h holds the same kernel as a template over storage type / frames per lane / weight layout (bf16 path, A/B
// variants).  Instantiated for (float, 4 frames, torch weights) that template computes bit-identical results but runs 10-23 %
// SLOWER in the plain and statistics flavours (same-process A/B, tools/ubench/ab_gc_r1.py: 69.5 vs 62.7 us, 93.1 vs 82.8,
// 69.2 vs 56.1, 41.6 vs 37.8 us over the four blocks; equal with LayerNorm on load): with the per-frame statistics of skip0 in
// arrays instead of float4 members hipcc no longer hoists their compares and subtractions out of the channel loop of the
// epilogue (32 v_cmp + 32 v_sub instead of 4 + packed adds, 33 s_nop instead of 6 in the CG = 8 instance).  The graded kernel
// therefore stays the tuned source; the template serves everything else.
// LNX: the main input carries a pending LayerNorm (deferred normalisation, nbasr.h) applied while loading;
// ln_s0.stats != nullptr: skip0 carries one (inside a cell both are the cell input, with the same statistics).
// STATS: the epilogue also emits this workgroup's partial LayerNorm statistics of y -- per frame the (mean, M2) over the
// 4 x CG channels of its four groups -- to `part` ([group quad][batch][2][ld]); stats_finalize_kernel merges the quads.
// This replaces the separate statistics pass over y when y is the last node of a cell.
template <int CG, int K, int D, bool LNX, bool STATS>
__global__ __launch_bounds__(256) void grouped_conv_f32_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
    const float* __restrict__ s0, const float* __restrict__ s1, const float* __restrict__ s2,
    float* __restrict__ y, int channels, int frames, int ld, int groups, const LnRef ln_x, const LnRef ln_s0,
    float* __restrict__ part)
{
    constexpr int LPAD = pad_left(K, D, 1);
    constexpr int SPAN = (K - 1) * D;            // taps reach frames [t - LPAD, t - LPAD + SPAN]
    constexpr int QL = (LPAD + 3) / 4;           // whole chunks left of the lane's own chunk
    constexpr int QR = (SPAN - LPAD + 3) / 4;    // whole chunks right of it
    constexpr int NCH = QL + 1 + QR;
    constexpr int BASE = 4 * QL - LPAD;          // window index of (r = 0, tap = 0)

    const int nq = ld >> 2;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int q = blockIdx.x * 64 + lane;
    // wave-uniform group index (scalar registers => s_load for weights and bias)
    const int g = __builtin_amdgcn_readfirstlane(blockIdx.y * 4 + wave);
    const int b = blockIdx.z;
    if (!STATS && g >= groups) return;              // with STATS every wave must reach the workgroup barrier below
    const bool active = q < nq && g < groups;

    // a surplus wave of the last group quad (STATS flavour, groups % 4 != 0) computes on the last group's weights and
    // stores nothing: every address below stays inside the tensors
    const int ga = g < groups ? g : groups - 1;
    const size_t row0 = (static_cast<size_t>(b) * channels + static_cast<size_t>(ga) * CG) * ld;
    const float* __restrict__ wg = w + static_cast<size_t>(ga) * (CG * CG * K);
    const float* __restrict__ bg = bias + ga * CG;

    float acc[CG][4];
#pragma unroll
    for (int co = 0; co < CG; ++co) {
        const float bv = bg[co];
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[co][r] = bv;
    }

    // per-frame LayerNorm statistics of the window (shared by all input channels), kept as frame PAIRS so that the
    // normalisation below is packed arithmetic: -mean, rstd and a 0/1 mask (rstd == 0 marks frames outside the utterance,
    // which must stay exactly 0).  (x + -mean) * rstd, fma(., gamma, beta), * mask rounds exactly like ln_apply.
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
            if (active && qq >= 0 && qq < nq) { m = mrow[qq]; r = rrow[qq]; }
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
            if (active && qq >= 0 && qq < nq) v = xrow[qq];
            xw[4 * c + 0] = v.x; xw[4 * c + 1] = v.y; xw[4 * c + 2] = v.z; xw[4 * c + 3] = v.w;
        }
        if (LNX) {
            const float gam = ln_x.gamma[ga * CG + ci], bet = ln_x.beta[ga * CG + ci];     // wave-uniform: scalar loads
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
                for (int r = 0; r < 4; ++r)
                    acc[co][r] = __builtin_fmaf(wv, xw[BASE + r + j * D], acc[co][r]);
            }
        }
    }

    if (!STATS && !active) return;
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
    for (int co = 0; co < CG; ++co) {
        const size_t off = row0 + static_cast<size_t>(co) * ld + t0;
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = relu_clamp(acc[co][r]);
        if (s0) {
            float4 v = *reinterpret_cast<const float4*>(s0 + off);
            if (ln_s0.stats) {
                const float gam = ln_s0.gamma[ga * CG + co], bet = ln_s0.beta[ga * CG + co];
                v.x = ln_apply(v.x, sm.x, sr.x, gam, bet); v.y = ln_apply(v.y, sm.y, sr.y, gam, bet);
                v.z = ln_apply(v.z, sm.z, sr.z, gam, bet); v.w = ln_apply(v.w, sm.w, sr.w, gam, bet);
            }
            o[0] += v.x; o[1] += v.y; o[2] += v.z; o[3] += v.w;
        }
        if (s1) { const float4 v = *reinterpret_cast<const float4*>(s1 + off); o[0] += v.x; o[1] += v.y; o[2] += v.z; o[3] += v.w; }
        if (s2) { const float4 v = *reinterpret_cast<const float4*>(s2 + off); o[0] += v.x; o[1] += v.y; o[2] += v.z; o[3] += v.w; }
        // keep the pitch columns frames..ld-1 at zero (layout invariant, nbasr.h); only the wave that holds the ragged chunk
        if (ragged) {
#pragma unroll
            for (int r = 0; r < 4; ++r) if (t0 + r >= frames) o[r] = 0.f;
        }
        typedef float f4v __attribute__((ext_vector_type(4)));
        __builtin_nontemporal_store(f4v{o[0], o[1], o[2], o[3]}, reinterpret_cast<f4v*>(y + off));
        if (STATS) {
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[co][r] = o[r];           // keep the final values for the statistics
        }
    }
    }
    if (STATS) {
        // per-lane (mean, M2) over this group's CG channels, exact two-pass in registers
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

template <int CG, int K, int D>
static int launch_grouped_f32(const GroupedArgs<float>& a, hipStream_t stream)
{
    const int nq = a.ld / 4;
    dim3 grid((nq + 63) / 64, (a.groups + 3) / 4, a.batch);
#define NBASR_LAUNCH_GROUPED(LNX, STATS)                                                                                  \
    hipLaunchKernelGGL((grouped_conv_f32_kernel<CG, K, D, LNX, STATS>), grid, dim3(256), 0, stream, a.x, a.w, a.bias, a.s0, a.s1, \
                       a.s2, a.y, a.channels, a.frames, a.ld, a.groups, a.ln_x, a.ln_s0, a.part)
    if (a.ln_x.stats) { if (a.part) NBASR_LAUNCH_GROUPED(true, true); else NBASR_LAUNCH_GROUPED(true, false); }
    else              { if (a.part) NBASR_LAUNCH_GROUPED(false, true); else NBASR_LAUNCH_GROUPED(false, false); }
#undef NBASR_LAUNCH_GROUPED
    return launch_status("nbasr_grouped_conv1d_node");
}

template <int CG>
static int dispatch_kd_f32(int kernel, int dilation, const GroupedArgs<float>& a, hipStream_t stream)
{
    if (kernel == 5 && dilation == 1) return launch_grouped_f32<CG, 5, 1>(a, stream);
    if (kernel == 5 && dilation == 2) return launch_grouped_f32<CG, 5, 2>(a, stream);
    if (kernel == 7 && dilation == 1) return launch_grouped_f32<CG, 7, 1>(a, stream);
    if (kernel == 7 && dilation == 2) return launch_grouped_f32<CG, 7, 2>(a, stream);
    set_error("nbasr_grouped_conv1d_node: unsupported (kernel=%d, dilation=%d); search space has k in {5,7}, d in {1,2}", kernel, dilation);
    return NBASR_EINVAL;
}

int grouped_conv_f32_base(const GroupedArgs<float>& a, int kernel, int dilation, hipStream_t stream)
{
    switch (a.channels / a.groups) {
        case 6:  return dispatch_kd_f32<6>(kernel, dilation, a, stream);
        case 8:  return dispatch_kd_f32<8>(kernel, dilation, a, stream);
        case 10: return dispatch_kd_f32<10>(kernel, dilation, a, stream);
        case 12: return dispatch_kd_f32<12>(kernel, dilation, a, stream);
        default:
            set_error("nbasr_grouped_conv1d_node: channels/groups=%d unsupported (model widths give 6, 8, 10, 12)", a.channels / a.groups);
            return NBASR_EINVAL;
    }
}

}  // namespace nbasr

using namespace nbasr;

extern "C" size_t nbasr_grouped_stats_workspace_bytes(int batch, int ld, int groups)
{
    if (batch <= 0 || ld <= 0 || groups <= 0) return 0;
    return static_cast<size_t>(groups) * batch * 2 * ld * sizeof(float);      // room for per-GROUP partials (fused cells of one group per workgroup); the node kernels use a quarter
}

// every flavour of the node op lands here: validation, then the variant's translation unit
static int grouped_node_impl(const char* what, const void* x, const float* w, const float* bias, const void* skip0, const void* skip1,
                             const void* skip2, void* y, int batch, int channels, int frames, int ld, int groups, int kernel,
                             int dilation, const nbasr_deferred_ln* ln, int ln_on_x, int ln_on_skip0, float* stats_ws, int dtype,
                             int variant, nbasr_stream_t stream)
{
    NBASR_REQUIRE(dtype == NBASR_F32 || dtype == NBASR_BF16, NBASR_EINVAL, "%s: dtype %d is neither NBASR_F32 nor NBASR_BF16", what, dtype);
    const bool alt2 = variant > 0 && (variant & ~(NBASR_GC_OSPLIT | NBASR_GC_PIPE)) == 0;       // output split and / or pipelined loads
    const bool ring = variant == NBASR_GC_RING;                                                 // windows staged through LDS by LDS-DMA
    NBASR_REQUIRE((variant >= 0 && variant <= (NBASR_GC_FPL8 | NBASR_GC_WPERM)) || ((alt2 || ring) && dtype == NBASR_F32), NBASR_EINVAL,
                  "%s: unknown variant %d (NBASR_GC_OSPLIT / NBASR_GC_PIPE / NBASR_GC_RING: fp32 only, not with the other bits)", what, variant);
    NBASR_REQUIRE(aligned16(stats_ws), NBASR_EALIGN, "%s: statistics buffers must be 16-byte aligned", what);
    NBASR_REQUIRE(batch >= 0 && channels > 0 && frames >= 0 && groups > 0 && channels % groups == 0, NBASR_EINVAL,
                  "%s: bad sizes batch=%d channels=%d frames=%d groups=%d", what, batch, channels, frames, groups);
    if (batch == 0 || ld == 0) return NBASR_OK;      // empty batch: nothing to do (empty tensors have NULL storage)
    NBASR_REQUIRE(x && w && bias && y, NBASR_ENULL, "%s: x, w, bias, y must be non-NULL", what);
    // a lane moves 16 bytes: 4 fp32 or 8 bf16 frames (8 fp32 frames as two accesses); rows are pitched to whole lanes
    const int pitch = (dtype == NBASR_BF16 || (variant & NBASR_GC_FPL8)) ? 8 : 4;
    NBASR_REQUIRE(ld >= frames && ld % pitch == 0, NBASR_EALIGN, "%s: ld=%d must be >= frames=%d and a multiple of %d", what, ld, frames, pitch);
    NBASR_REQUIRE(aligned16(x) && aligned16(y) && aligned16(skip0) && aligned16(skip1) && aligned16(skip2), NBASR_EALIGN,
                  "%s: activation pointers must be 16-byte aligned", what);
    NBASR_REQUIRE(static_cast<long long>(batch) * (ld / 4) < (1ll << 31) && batch <= 65535, NBASR_EINVAL, "%s: batch * ld too large", what);
    const bool any_ln = ln && (ln_on_x || (ln_on_skip0 && skip0));
    NBASR_REQUIRE(!any_ln || (ln->stats && ln->gamma && ln->beta && aligned16(ln->stats)), NBASR_ENULL,
                  "%s: deferred LayerNorm needs stats (16-byte aligned), gamma and beta", what);
    const LnRef lx = ln_ref(ln, ln_on_x != 0), ls = ln_ref(ln, ln_on_skip0 != 0 && skip0 != nullptr);
    hipStream_t s = as_stream(stream);
    if (dtype == NBASR_F32) {
        GroupedArgs<float> a{static_cast<const float*>(x), w, bias, static_cast<const float*>(skip0), static_cast<const float*>(skip1),
                             static_cast<const float*>(skip2), static_cast<float*>(y), batch, channels, frames, ld, groups, lx, ls, stats_ws};
        if (ring) return grouped_conv_f32_ring(variant, a, kernel, dilation, s);
        if (alt2) return grouped_conv_f32_osplit(variant, a, kernel, dilation, s);
        return variant == 0 ? grouped_conv_f32_base(a, kernel, dilation, s) : grouped_conv_f32_alt(variant, a, kernel, dilation, s);
    }
    GroupedArgs<bf16_t> a{static_cast<const bf16_t*>(x), w, bias, static_cast<const bf16_t*>(skip0), static_cast<const bf16_t*>(skip1),
                          static_cast<const bf16_t*>(skip2), static_cast<bf16_t*>(y), batch, channels, frames, ld, groups, lx, ls, stats_ws};
    return grouped_conv_bf16(variant, a, kernel, dilation, s);
}

extern "C" int nbasr_grouped_conv1d_node(const void* x, const float* w, const float* bias, const void* skip0, const void* skip1,
                                         const void* skip2, void* y, int batch, int channels, int frames, int ld, int groups,
                                         int kernel, int dilation, const nbasr_deferred_ln* ln, int ln_on_x, int ln_on_skip0,
                                         float* stats_ws, int dtype, int variant, nbasr_stream_t stream)
{
    clear_error();
    return grouped_node_impl("nbasr_grouped_conv1d_node", x, w, bias, skip0, skip1, skip2, y, batch, channels, frames, ld, groups, kernel,
                             dilation, ln, ln_on_x, ln_on_skip0, stats_ws, dtype, variant, stream);
}

extern "C" int nbasr_pack_grouped_weights(const float* w, float* packed, int channels, int groups, int kernel, nbasr_stream_t stream)
{
    clear_error();
    NBASR_REQUIRE(channels > 0 && groups > 0 && channels % groups == 0 && kernel > 0, NBASR_EINVAL, "nbasr_pack_grouped_weights: bad sizes");
    NBASR_REQUIRE(w && packed, NBASR_ENULL, "nbasr_pack_grouped_weights: NULL pointer");
    const int n = channels * (channels / groups) * kernel;
    hipLaunchKernelGGL(pack_grouped_weights_kernel, dim3((n + 255) / 256), dim3(256), 0, as_stream(stream), w, packed, channels,
                       channels / groups, kernel);
    return launch_status("nbasr_pack_grouped_weights");
}

extern "C" int nbasr_grouped_stats_finalize(const float* stats_ws, float* stats_out, int batch, int channels, int frames, int ld,
                                            int groups, int groups_per_part, float eps, nbasr_stream_t stream)
{
    clear_error();
    NBASR_REQUIRE(batch >= 0 && channels > 0 && groups > 0 && channels % groups == 0 && frames >= 0 && ld >= frames, NBASR_EINVAL,
                  "nbasr_grouped_stats_finalize: bad sizes");
    NBASR_REQUIRE(groups_per_part >= 1, NBASR_EINVAL, "nbasr_grouped_stats_finalize: groups_per_part=%d (4, 2 or 1 for grouped convolutions, the row tile for a dense one)", groups_per_part);
    if (batch == 0 || ld == 0) return NBASR_OK;
    NBASR_REQUIRE(stats_ws && stats_out, NBASR_ENULL, "nbasr_grouped_stats_finalize: NULL pointer");
    NBASR_REQUIRE(ld % 4 == 0 && aligned16(stats_ws), NBASR_EALIGN, "nbasr_grouped_stats_finalize: ld=%d must be a multiple of 4 and stats_ws 16-byte aligned", ld);
    NBASR_REQUIRE((groups + groups_per_part - 1) / groups_per_part <= SF_LANES * SF_MAX_PER_LANE, NBASR_EINVAL,
                  "nbasr_grouped_stats_finalize: %d groups in parts of %d are more than %d partials per frame", groups, groups_per_part, SF_LANES * SF_MAX_PER_LANE);
    hipLaunchKernelGGL(stats_finalize_kernel, dim3((ld + SF_FRAMES - 1) / SF_FRAMES, batch), dim3(SF_QUADS * SF_LANES), 0, as_stream(stream), stats_ws, stats_out,
                       batch, frames, ld, groups, channels / groups, groups_per_part, eps);
    return launch_status("nbasr_grouped_stats_finalize");
}

extern "C" int nbasr_skip_sum(const void* skip0, const void* skip1, const void* skip2, void* y, int batch,
                                int channels, int frames, int ld, const nbasr_deferred_ln* ln, int ln_on_skip0, int dtype,
                                nbasr_stream_t stream)
{
    clear_error();
    NBASR_REQUIRE(dtype == NBASR_F32 || dtype == NBASR_BF16, NBASR_EINVAL, "nbasr_skip_sum: unknown dtype %d", dtype);
    NBASR_REQUIRE(batch >= 0 && channels > 0 && frames >= 0, NBASR_EINVAL, "nbasr_skip_sum: bad sizes");
    if (batch == 0 || ld == 0) return NBASR_OK;
    NBASR_REQUIRE(y, NBASR_ENULL, "nbasr_skip_sum: y must be non-NULL");
    const int fr = dtype == NBASR_BF16 ? 8 : 4;
    NBASR_REQUIRE(ld >= frames && ld % fr == 0, NBASR_EALIGN, "nbasr_skip_sum: ld=%d must be >= frames and a multiple of %d", ld, fr);
    NBASR_REQUIRE(aligned16(y) && aligned16(skip0) && aligned16(skip1) && aligned16(skip2), NBASR_EALIGN,
                  "nbasr_skip_sum: pointers must be 16-byte aligned");
    const bool use_ln = ln && ln_on_skip0 && skip0;
    NBASR_REQUIRE(!use_ln || (ln->stats && ln->gamma && ln->beta && aligned16(ln->stats)), NBASR_ENULL,
                  "nbasr_skip_sum: deferred LayerNorm needs stats (16-byte aligned), gamma and beta");
    const size_t nchunks = static_cast<size_t>(batch) * channels * ld / fr;
    const unsigned blocks = static_cast<unsigned>(nchunks / 256 + 1 < 4096 ? nchunks / 256 + 1 : 4096);
    if (dtype == NBASR_BF16)
        hipLaunchKernelGGL(skip_sum_kernel<bf16_t>, dim3(blocks), dim3(256), 0, as_stream(stream), static_cast<const bf16_t*>(skip0),
                           static_cast<const bf16_t*>(skip1), static_cast<const bf16_t*>(skip2), static_cast<bf16_t*>(y), nchunks, channels,
                           ld / fr, ln_ref(ln, use_ln));
    else
        hipLaunchKernelGGL(skip_sum_kernel<float>, dim3(blocks), dim3(256), 0, as_stream(stream), static_cast<const float*>(skip0),
                           static_cast<const float*>(skip1), static_cast<const float*>(skip2), static_cast<float*>(y), nchunks, channels,
                           ld / fr, ln_ref(ln, use_ln));
    return launch_status("nbasr_skip_sum");
}

extern "C" int nbasr_repitch(const void* src, void* dst, int rows, int frames, int ld_src, int ld_dst, int dtype,
                               nbasr_stream_t stream)
{
    clear_error();
    NBASR_REQUIRE(dtype == NBASR_F32 || dtype == NBASR_BF16, NBASR_EINVAL, "nbasr_repitch: unknown dtype %d", dtype);
    NBASR_REQUIRE(rows >= 0 && frames >= 0 && ld_src >= frames && ld_dst >= frames, NBASR_EINVAL, "nbasr_repitch: bad sizes");
    if (rows == 0 || ld_dst == 0) return NBASR_OK;
    NBASR_REQUIRE(src && dst, NBASR_ENULL, "nbasr_repitch: NULL pointer");
    const dim3 grid((ld_dst + 255) / 256, rows < 65535 ? rows : 65535);
    if (dtype == NBASR_BF16)
        hipLaunchKernelGGL(repitch_kernel<unsigned short>, grid, dim3(256), 0, as_stream(stream), static_cast<const unsigned short*>(src),
                           static_cast<unsigned short*>(dst), rows, frames, ld_src, ld_dst);
    else
        hipLaunchKernelGGL(repitch_kernel<float>, grid, dim3(256), 0, as_stream(stream), static_cast<const float*>(src),
                           static_cast<float*>(dst), rows, frames, ld_src, ld_dst);
    return launch_status("nbasr_repitch");
}

