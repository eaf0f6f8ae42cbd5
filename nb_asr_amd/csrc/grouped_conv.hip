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

// merge the per-quad partial statistics: stats (batch, 2, ld) <- (mean, rstd) per frame, 0 in the pitch columns
__global__ __launch_bounds__(256) void stats_finalize_kernel(const float* __restrict__ part, float* __restrict__ stats,
                                                             int batch, int frames, int ld, int groups, int cg, float eps)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (t >= ld) return;
    float* srow = stats + static_cast<size_t>(b) * 2 * ld;
    if (t >= frames) { srow[t] = 0.f; srow[ld + t] = 0.f; return; }
    const int nquads = (groups + 3) / 4;
    float cnt = 0.f, mean = 0.f, m2 = 0.f;
    // the partials are loaded five at a time BEFORE they are merged: the merge is a serial chain, the loads need not be
    // (25 partials per frame at 100 groups: 5 round trips to memory instead of 25; same merge order, same result)
    for (int k0 = 0; k0 < nquads; k0 += 5) {
        float pm[5], pq[5];
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            const int k = min(k0 + u, nquads - 1);
            const float* prow = part + (static_cast<size_t>(k) * batch + b) * 2 * ld;
            pm[u] = prow[t];
            pq[u] = prow[ld + t];
        }
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            const int k = k0 + u;
            if (k >= nquads) break;
            const float nb = static_cast<float>(cg * min(4, groups - 4 * k));
            const float tot = cnt + nb;
            const float delta = pm[u] - mean;
            mean += delta * (nb / tot);
            m2 += pq[u] + delta * delta * (cnt * nb / tot);
            cnt = tot;
        }
    }
    srow[t] = mean;
    srow[ld + t] = 1.0f / sqrtf(m2 / cnt + eps);
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

int grouped_conv_f32_base(const GroupedArgs<float>& a, int kernel, int dilation, hipStream_t stream)
{
    return grouped_conv_variant<float, 4, false>(a, kernel, dilation, stream);
}

}  // namespace nbasr

using namespace nbasr;

extern "C" size_t nbasr_grouped_stats_workspace_bytes(int batch, int ld, int groups)
{
    if (batch <= 0 || ld <= 0 || groups <= 0) return 0;
    return static_cast<size_t>((groups + 3) / 4) * batch * 2 * ld * sizeof(float);
}

// every flavour of the node op lands here: validation, then the variant's translation unit
static int grouped_node_impl(const char* what, const void* x, const float* w, const float* bias, const void* skip0, const void* skip1,
                             const void* skip2, void* y, int batch, int channels, int frames, int ld, int groups, int kernel,
                             int dilation, const nbasr_deferred_ln* ln, int ln_on_x, int ln_on_skip0, float* stats_ws, int dtype,
                             int variant /* by value: the KEEP bit is peeled off below */, nbasr_stream_t stream)
{
    NBASR_REQUIRE(dtype == NBASR_F32 || dtype == NBASR_BF16, NBASR_EINVAL, "%s: dtype %d is neither NBASR_F32 nor NBASR_BF16", what, dtype);
    const int keep = (variant & NBASR_GC_KEEP) ? 1 : 0;
    variant &= ~NBASR_GC_KEEP;
    NBASR_REQUIRE((variant >= 0 && variant <= (NBASR_GC_FPL8 | NBASR_GC_WPERM)) || (variant == NBASR_GC_FPL2 && dtype == NBASR_F32), NBASR_EINVAL,
                  "%s: unknown variant %d", what, variant);
    NBASR_REQUIRE(aligned16(stats_ws), NBASR_EALIGN, "%s: statistics buffers must be 16-byte aligned", what);
    NBASR_REQUIRE(batch >= 0 && channels > 0 && frames >= 0 && groups > 0 && channels % groups == 0, NBASR_EINVAL,
                  "%s: bad sizes batch=%d channels=%d frames=%d groups=%d", what, batch, channels, frames, groups);
    if (batch == 0 || ld == 0) return NBASR_OK;      // empty batch: nothing to do (empty tensors have NULL storage)
    NBASR_REQUIRE(x && w && bias && y, NBASR_ENULL, "%s: x, w, bias, y must be non-NULL", what);
    // a lane moves 16 bytes: 4 fp32 or 8 bf16 frames (8 fp32 frames as two accesses); rows are pitched to whole lanes
    const int pitch = (dtype == NBASR_BF16 || (variant != NBASR_GC_FPL2 && (variant & NBASR_GC_FPL8))) ? 8 : 4;
    NBASR_REQUIRE(ld >= frames && ld % pitch == 0, NBASR_EALIGN, "%s: ld=%d must be >= frames=%d and a multiple of %d", what, ld, frames, pitch);
    NBASR_REQUIRE(aligned16(x) && aligned16(y) && aligned16(skip0) && aligned16(skip1) && aligned16(skip2), NBASR_EALIGN,
                  "%s: activation pointers must be 16-byte aligned", what);
    NBASR_REQUIRE(static_cast<long long>(batch) * (ld / 4) < (1ll << 31), NBASR_EINVAL, "%s: batch * ld too large", what);
    const bool any_ln = ln && (ln_on_x || (ln_on_skip0 && skip0));
    NBASR_REQUIRE(!any_ln || (ln->stats && ln->gamma && ln->beta && aligned16(ln->stats)), NBASR_ENULL,
                  "%s: deferred LayerNorm needs stats (16-byte aligned), gamma and beta", what);
    const LnRef lx = ln_ref(ln, ln_on_x != 0), ls = ln_ref(ln, ln_on_skip0 != 0 && skip0 != nullptr);
    hipStream_t s = as_stream(stream);
    if (dtype == NBASR_F32) {
        GroupedArgs<float> a{static_cast<const float*>(x), w, bias, static_cast<const float*>(skip0), static_cast<const float*>(skip1),
                             static_cast<const float*>(skip2), static_cast<float*>(y), batch, channels, frames, ld, groups, lx, ls, stats_ws, keep};
        if (variant == NBASR_GC_FPL2) return grouped_conv_f32_fpl2(a, kernel, dilation, s);
        return variant == 0 ? grouped_conv_f32_base(a, kernel, dilation, s) : grouped_conv_f32_alt(variant, a, kernel, dilation, s);
    }
    GroupedArgs<bf16_t> a{static_cast<const bf16_t*>(x), w, bias, static_cast<const bf16_t*>(skip0), static_cast<const bf16_t*>(skip1),
                          static_cast<const bf16_t*>(skip2), static_cast<bf16_t*>(y), batch, channels, frames, ld, groups, lx, ls, stats_ws, keep};
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

extern "C" int nbasr_grouped_conv1d_fused_stats(const float* x, const float* w, const float* bias, const float* skip0,
                                                const float* skip1, const float* skip2, float* y, int batch, int channels,
                                                int frames, int ld, int groups, int kernel, int dilation,
                                                const nbasr_deferred_ln* ln, int ln_on_x, int ln_on_skip0,
                                                float* stats_out, float* stats_ws, float eps, nbasr_stream_t stream)
{
    clear_error();
    NBASR_REQUIRE(!stats_out || stats_ws, NBASR_ENULL,
                  "nbasr_grouped_conv1d_fused_stats: stats_out needs the partial-statistics workspace stats_ws");
    NBASR_REQUIRE(aligned16(stats_out), NBASR_EALIGN, "nbasr_grouped_conv1d_fused_stats: statistics buffers must be 16-byte aligned");
    const int rc = grouped_node_impl("nbasr_grouped_conv1d_fused", x, w, bias, skip0, skip1, skip2, y, batch, channels, frames, ld, groups,
                                     kernel, dilation, ln, ln_on_x, ln_on_skip0, stats_ws, NBASR_F32, 0, stream);
    if (rc != NBASR_OK || !stats_out || batch == 0 || ld == 0) return rc;   // stats_ws alone: partials only, merge later with nbasr_grouped_stats_finalize
    return nbasr_grouped_stats_finalize(stats_ws, stats_out, batch, channels, frames, ld, groups, eps, stream);
}

extern "C" int nbasr_grouped_stats_finalize(const float* stats_ws, float* stats_out, int batch, int channels, int frames, int ld,
                                            int groups, float eps, nbasr_stream_t stream)
{
    clear_error();
    NBASR_REQUIRE(batch >= 0 && channels > 0 && groups > 0 && channels % groups == 0 && frames >= 0 && ld >= frames, NBASR_EINVAL,
                  "nbasr_grouped_stats_finalize: bad sizes");
    if (batch == 0 || ld == 0) return NBASR_OK;
    NBASR_REQUIRE(stats_ws && stats_out, NBASR_ENULL, "nbasr_grouped_stats_finalize: NULL pointer");
    hipLaunchKernelGGL(stats_finalize_kernel, dim3((ld + 255) / 256, batch), dim3(256), 0, as_stream(stream), stats_ws, stats_out,
                       batch, frames, ld, groups, channels / groups, eps);
    return launch_status("nbasr_grouped_stats_finalize");
}

extern "C" int nbasr_grouped_conv1d_fused_ln(const float* x, const float* w, const float* bias, const float* skip0,
                                             const float* skip1, const float* skip2, float* y, int batch, int channels,
                                             int frames, int ld, int groups, int kernel, int dilation,
                                             const nbasr_deferred_ln* ln, int ln_on_x, int ln_on_skip0, nbasr_stream_t stream)
{
    return nbasr_grouped_conv1d_fused_stats(x, w, bias, skip0, skip1, skip2, y, batch, channels, frames, ld, groups, kernel,
                                            dilation, ln, ln_on_x, ln_on_skip0, nullptr, nullptr, 0.f, stream);
}

extern "C" int nbasr_grouped_conv1d_fused(const float* x, const float* w, const float* bias, const float* skip0,
                                          const float* skip1, const float* skip2, float* y, int batch, int channels,
                                          int frames, int ld, int groups, int kernel, int dilation,
                                          nbasr_stream_t stream)
{
    return nbasr_grouped_conv1d_fused_ln(x, w, bias, skip0, skip1, skip2, y, batch, channels, frames, ld, groups, kernel,
                                         dilation, nullptr, 0, 0, stream);
}

extern "C" int nbasr_skip_sum_v(const void* skip0, const void* skip1, const void* skip2, void* y, int batch,
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
                  "nbasr_skip_sum_ln: deferred LayerNorm needs stats (16-byte aligned), gamma and beta");
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

extern "C" int nbasr_skip_sum_ln(const float* skip0, const float* skip1, const float* skip2, float* y, int batch,
                                 int channels, int frames, int ld, const nbasr_deferred_ln* ln, int ln_on_skip0,
                                 nbasr_stream_t stream)
{
    return nbasr_skip_sum_v(skip0, skip1, skip2, y, batch, channels, frames, ld, ln, ln_on_skip0, NBASR_F32, stream);
}

extern "C" int nbasr_skip_sum(const float* skip0, const float* skip1, const float* skip2, float* y, int batch,
                              int channels, int frames, int ld, nbasr_stream_t stream)
{
    return nbasr_skip_sum_ln(skip0, skip1, skip2, y, batch, channels, frames, ld, nullptr, 0, stream);
}

extern "C" int nbasr_repitch_v(const void* src, void* dst, int rows, int frames, int ld_src, int ld_dst, int dtype,
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

extern "C" int nbasr_repitch(const float* src, float* dst, int rows, int frames, int ld_src, int ld_dst,
                             nbasr_stream_t stream)
{
    return nbasr_repitch_v(src, dst, rows, frames, ld_src, ld_dst, NBASR_F32, stream);
}
