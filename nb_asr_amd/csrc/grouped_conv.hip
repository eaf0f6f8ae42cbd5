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
#include "common.h"

namespace nbasr {

template <int CG, int K, int D>
__global__ __launch_bounds__(256) void grouped_conv_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
    const float* __restrict__ s0, const float* __restrict__ s1, const float* __restrict__ s2,
    float* __restrict__ y, int channels, int frames, int ld, int groups)
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
    if (g >= groups) return;
    const bool active = q < nq;

    const size_t row0 = (static_cast<size_t>(b) * channels + static_cast<size_t>(g) * CG) * ld;
    const float* __restrict__ wg = w + static_cast<size_t>(g) * (CG * CG * K);
    const float* __restrict__ bg = bias + g * CG;

    float acc[CG][4];
#pragma unroll
    for (int co = 0; co < CG; ++co) {
        const float bv = bg[co];
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[co][r] = bv;
    }

#pragma unroll 2
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

    if (!active) return;
    const int t0 = q * 4;
#pragma unroll
    for (int co = 0; co < CG; ++co) {
        const size_t off = row0 + static_cast<size_t>(co) * ld + t0;
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = relu_clamp(acc[co][r]);
        if (s0) { const float4 v = *reinterpret_cast<const float4*>(s0 + off); o[0] += v.x; o[1] += v.y; o[2] += v.z; o[3] += v.w; }
        if (s1) { const float4 v = *reinterpret_cast<const float4*>(s1 + off); o[0] += v.x; o[1] += v.y; o[2] += v.z; o[3] += v.w; }
        if (s2) { const float4 v = *reinterpret_cast<const float4*>(s2 + off); o[0] += v.x; o[1] += v.y; o[2] += v.z; o[3] += v.w; }
        // keep the pitch columns frames..ld-1 at zero (layout invariant, nbasr.h)
#pragma unroll
        for (int r = 0; r < 4; ++r) if (t0 + r >= frames) o[r] = 0.f;
        *reinterpret_cast<float4*>(y + off) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// y = 0 + skip0 + skip1 + skip2 for a node whose main op is `zero` (reference ops.py:67-68)
__global__ __launch_bounds__(256) void skip_sum_kernel(
    const float* __restrict__ s0, const float* __restrict__ s1, const float* __restrict__ s2,
    float* __restrict__ y, size_t n4)
{
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (s0) { const float4 v = reinterpret_cast<const float4*>(s0)[i]; o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w; }
        if (s1) { const float4 v = reinterpret_cast<const float4*>(s1)[i]; o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w; }
        if (s2) { const float4 v = reinterpret_cast<const float4*>(s2)[i]; o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w; }
        reinterpret_cast<float4*>(y)[i] = o;
    }
}

__global__ __launch_bounds__(256) void repitch_kernel(
    const float* __restrict__ src, float* __restrict__ dst, int rows, int frames, int ld_src, int ld_dst)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ld_dst) return;
    for (int r = blockIdx.y; r < rows; r += gridDim.y)
        dst[static_cast<size_t>(r) * ld_dst + t] = (t < frames) ? src[static_cast<size_t>(r) * ld_src + t] : 0.f;
}

template <int CG, int K, int D>
static int launch_grouped(const float* x, const float* w, const float* bias, const float* s0, const float* s1,
                          const float* s2, float* y, int batch, int channels, int frames, int ld, int groups,
                          hipStream_t stream)
{
    const int nq = ld / 4;
    dim3 grid((nq + 63) / 64, (groups + 3) / 4, batch);
    hipLaunchKernelGGL((grouped_conv_kernel<CG, K, D>), grid, dim3(256), 0, stream,
                       x, w, bias, s0, s1, s2, y, channels, frames, ld, groups);
    return launch_status("nbasr_grouped_conv1d_fused");
}

template <int CG>
static int dispatch_kd(int kernel, int dilation, const float* x, const float* w, const float* bias, const float* s0,
                       const float* s1, const float* s2, float* y, int batch, int channels, int frames, int ld,
                       int groups, hipStream_t stream)
{
    if (kernel == 5 && dilation == 1) return launch_grouped<CG, 5, 1>(x, w, bias, s0, s1, s2, y, batch, channels, frames, ld, groups, stream);
    if (kernel == 5 && dilation == 2) return launch_grouped<CG, 5, 2>(x, w, bias, s0, s1, s2, y, batch, channels, frames, ld, groups, stream);
    if (kernel == 7 && dilation == 1) return launch_grouped<CG, 7, 1>(x, w, bias, s0, s1, s2, y, batch, channels, frames, ld, groups, stream);
    if (kernel == 7 && dilation == 2) return launch_grouped<CG, 7, 2>(x, w, bias, s0, s1, s2, y, batch, channels, frames, ld, groups, stream);
    set_error("nbasr_grouped_conv1d_fused: unsupported (kernel=%d, dilation=%d); search space has k in {5,7}, d in {1,2}", kernel, dilation);
    return NBASR_EINVAL;
}

}  // namespace nbasr

using namespace nbasr;

extern "C" int nbasr_grouped_conv1d_fused(const float* x, const float* w, const float* bias, const float* skip0,
                                          const float* skip1, const float* skip2, float* y, int batch, int channels,
                                          int frames, int ld, int groups, int kernel, int dilation,
                                          nbasr_stream_t stream)
{
    clear_error();
    NBASR_REQUIRE(batch >= 0 && channels > 0 && frames >= 0 && groups > 0 && channels % groups == 0, NBASR_EINVAL,
                  "nbasr_grouped_conv1d_fused: bad sizes batch=%d channels=%d frames=%d groups=%d", batch, channels, frames, groups);
    if (batch == 0 || ld == 0) return NBASR_OK;      // empty batch: nothing to do (empty tensors have NULL storage)
    NBASR_REQUIRE(x && w && bias && y, NBASR_ENULL, "nbasr_grouped_conv1d_fused: x, w, bias, y must be non-NULL");
    NBASR_REQUIRE(ld >= frames && ld % 4 == 0, NBASR_EALIGN, "nbasr_grouped_conv1d_fused: ld=%d must be >= frames=%d and a multiple of 4", ld, frames);
    NBASR_REQUIRE(aligned16(x) && aligned16(y) && aligned16(skip0) && aligned16(skip1) && aligned16(skip2), NBASR_EALIGN,
                  "nbasr_grouped_conv1d_fused: activation pointers must be 16-byte aligned");
    NBASR_REQUIRE(batch <= 65535, NBASR_EINVAL, "nbasr_grouped_conv1d_fused: batch %d > 65535", batch);
    hipStream_t s = as_stream(stream);
    switch (channels / groups) {
        case 6:  return dispatch_kd<6>(kernel, dilation, x, w, bias, skip0, skip1, skip2, y, batch, channels, frames, ld, groups, s);
        case 8:  return dispatch_kd<8>(kernel, dilation, x, w, bias, skip0, skip1, skip2, y, batch, channels, frames, ld, groups, s);
        case 10: return dispatch_kd<10>(kernel, dilation, x, w, bias, skip0, skip1, skip2, y, batch, channels, frames, ld, groups, s);
        case 12: return dispatch_kd<12>(kernel, dilation, x, w, bias, skip0, skip1, skip2, y, batch, channels, frames, ld, groups, s);
        default:
            set_error("nbasr_grouped_conv1d_fused: channels/groups=%d unsupported (model widths give 6, 8, 10, 12)", channels / groups);
            return NBASR_EINVAL;
    }
}

extern "C" int nbasr_skip_sum(const float* skip0, const float* skip1, const float* skip2, float* y, int batch,
                              int channels, int frames, int ld, nbasr_stream_t stream)
{
    clear_error();
    NBASR_REQUIRE(batch >= 0 && channels > 0 && frames >= 0, NBASR_EINVAL, "nbasr_skip_sum: bad sizes");
    if (batch == 0 || ld == 0) return NBASR_OK;
    NBASR_REQUIRE(y, NBASR_ENULL, "nbasr_skip_sum: y must be non-NULL");
    NBASR_REQUIRE(ld >= frames && ld % 4 == 0, NBASR_EALIGN, "nbasr_skip_sum: ld=%d must be >= frames and a multiple of 4", ld);
    NBASR_REQUIRE(aligned16(y) && aligned16(skip0) && aligned16(skip1) && aligned16(skip2), NBASR_EALIGN,
                  "nbasr_skip_sum: pointers must be 16-byte aligned");
    const size_t n4 = static_cast<size_t>(batch) * channels * ld / 4;
    if (n4 == 0) return NBASR_OK;
    const unsigned blocks = static_cast<unsigned>(n4 / 256 + 1 < 4096 ? n4 / 256 + 1 : 4096);
    hipLaunchKernelGGL(skip_sum_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), skip0, skip1, skip2, y, n4);
    return launch_status("nbasr_skip_sum");
}

extern "C" int nbasr_repitch(const float* src, float* dst, int rows, int frames, int ld_src, int ld_dst,
                             nbasr_stream_t stream)
{
    clear_error();
    NBASR_REQUIRE(rows >= 0 && frames >= 0 && ld_src >= frames && ld_dst >= frames, NBASR_EINVAL, "nbasr_repitch: bad sizes");
    if (rows == 0 || ld_dst == 0) return NBASR_OK;
    NBASR_REQUIRE(src && dst, NBASR_ENULL, "nbasr_repitch: NULL pointer");
    if (rows == 0 || ld_dst == 0) return NBASR_OK;
    hipLaunchKernelGGL(repitch_kernel, dim3((ld_dst + 255) / 256, rows < 65535 ? rows : 65535), dim3(256), 0, as_stream(stream),
                       src, dst, rows, frames, ld_src, ld_dst);
    return launch_status("nbasr_repitch");
}
