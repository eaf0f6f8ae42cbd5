// LayerNorm over the CHANNEL dimension of a (batch, channels, frames) tensor, frames contiguous
// (reference model.py:92 + 125-128 after each downsample conv, model.py:46-47 + 55-58 at the end of
// each cell; eps = 1e-3, biased variance, affine).  The reference permutes to (B,T,C), normalises and
// permutes back, leaving a strided view; here the reduction dimension is simply the SLOW dimension,
// so no lane ever needs a cross-lane reduction for the statistics of its own frames.
//
// Mapping: a 256-thread workgroup owns 64 consecutive frames of one utterance as 16 lanes x 16-byte
// chunks (256-byte contiguous row segments) times 16 channel rows in flight; channels are strided
// over the 16 row slots.  Pass 1 accumulates shifted sums per lane (shift = first sample, so the
// single pass is cancellation-safe), the 16 row partials are merged with Chan's parallel-variance
// formula through LDS, pass 2 re-reads the tile (L2 / Infinity-Cache warm) and writes the result.
// HBM-bound: 2 reads + 1 write of the tensor (algorithmic minimum 1 read + 1 write).
#include "common.h"

namespace nbasr {

constexpr int LN_ROWS = 16;   // channel rows in flight per workgroup
constexpr int LN_QS = 16;     // 16-byte chunks (4 frames each) per row segment

// Statistics of one 64-frame tile: fills s_mu / s_rstd (LDS) for the tile's frame columns.  Called by all 256 threads.
__device__ __forceinline__ void tile_statistics(const float* x, int channels, int ld, float eps, size_t base, bool active,
                                                int row, int ql, float (*s_mean)[LN_QS * 4], float (*s_m2)[LN_QS * 4],
                                                float* s_cnt, float* s_mu, float* s_rstd)
{
    // per-lane shifted sums over this lane's channel subset
    float shift[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    int n = 0;
    if (active) {
        for (int c = row; c < channels; c += LN_ROWS) {
            const float4 v = *reinterpret_cast<const float4*>(x + base + static_cast<size_t>(c) * ld);
            const float e[4] = {v.x, v.y, v.z, v.w};
            if (n == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) shift[r] = e[r];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float d = e[r] - shift[r];
                s1[r] += d;
                s2[r] = __builtin_fmaf(d, d, s2[r]);
            }
            ++n;
        }
    }
    const float fn = static_cast<float>(n);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float dm = n ? s1[r] / fn : 0.f;
        s_mean[row][ql * 4 + r] = shift[r] + dm;
        s_m2[row][ql * 4 + r] = n ? fmaxf(s2[r] - dm * s1[r], 0.f) : 0.f;
    }
    if (ql == 0) s_cnt[row] = fn;
    __syncthreads();

    // merge the 16 row partials (Chan's parallel-variance formula), one thread per frame column
    if (threadIdx.x < LN_QS * 4) {
        const int col = threadIdx.x;
        float cnt = 0.f, mean = 0.f, m2 = 0.f;
#pragma unroll
        for (int k = 0; k < LN_ROWS; ++k) {
            const float nb = s_cnt[k];
            if (nb > 0.f) {
                const float tot = cnt + nb;
                const float delta = s_mean[k][col] - mean;
                mean += delta * (nb / tot);
                m2 += s_m2[k][col] + delta * delta * (cnt * nb / tot);
                cnt = tot;
            }
        }
        s_mu[col] = mean;
        s_rstd[col] = 1.0f / sqrtf(m2 / fmaxf(cnt, 1.f) + eps);
    }
    __syncthreads();
}

// ABSMAX: also fold max|y| of each utterance into absmax[b] (float bits, non-negative floats order like unsigned ints);
// the 2-way fp16 dense conv (gemm_conv_split.hip) scales its input by a power of two derived from it
template <bool ABSMAX>
__global__ __launch_bounds__(256) void layernorm_channels_kernel(
    const float* x, const float* __restrict__ gamma, const float* __restrict__ beta,
    float* y, int channels, int frames, int ld, float eps, unsigned* __restrict__ absmax)   // x may alias y (in-place)
{
    __shared__ float s_mean[LN_ROWS][LN_QS * 4];
    __shared__ float s_m2[LN_ROWS][LN_QS * 4];
    __shared__ float s_cnt[LN_ROWS];
    __shared__ float s_mu[LN_QS * 4];
    __shared__ float s_rstd[LN_QS * 4];

    const int ql = threadIdx.x & (LN_QS - 1);
    const int row = threadIdx.x / LN_QS;
    const int nq = ld >> 2;
    const int q = blockIdx.x * LN_QS + ql;
    const int b = blockIdx.y;
    const bool active = q < nq;
    const size_t base = static_cast<size_t>(b) * channels * ld + static_cast<size_t>(q) * 4;

    tile_statistics(x, channels, ld, eps, base, active, row, ql, s_mean, s_m2, s_cnt, s_mu, s_rstd);

    // pass 2: normalise, scale, shift; keep the pitch columns at zero
    if (!active && !ABSMAX) return;
    float mu[4], rs[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { mu[r] = s_mu[ql * 4 + r]; rs[r] = s_rstd[ql * 4 + r]; }
    const int t0 = q * 4;
    float amax = 0.f;
    for (int c = row; active && c < channels; c += LN_ROWS) {
        const size_t off = base + static_cast<size_t>(c) * ld;
        const float4 v = *reinterpret_cast<const float4*>(x + off);
        const float g = gamma[c], bt = beta[c];
        float o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            o[r] = (o[r] - mu[r]) * rs[r] * g + bt;
            if (t0 + r >= frames) o[r] = 0.f;
            if (ABSMAX) amax = fmaxf(amax, fabsf(o[r]));
        }
        *reinterpret_cast<float4*>(y + off) = make_float4(o[0], o[1], o[2], o[3]);
    }
    if (ABSMAX) {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) amax = fmaxf(amax, __shfl_xor(amax, m));
        if ((threadIdx.x & 63) == 0) atomicMax(absmax + b, __float_as_uint(amax));
    }
}

// deferred LayerNorm: ONE read pass, per-frame (mean, rstd) only; consumers normalise while loading
__global__ __launch_bounds__(256) void channel_stats_kernel(const float* __restrict__ x, float* __restrict__ stats,
                                                            int channels, int frames, int ld, float eps)
{
    __shared__ float s_mean[LN_ROWS][LN_QS * 4];
    __shared__ float s_m2[LN_ROWS][LN_QS * 4];
    __shared__ float s_cnt[LN_ROWS];
    __shared__ float s_mu[LN_QS * 4];
    __shared__ float s_rstd[LN_QS * 4];

    const int ql = threadIdx.x & (LN_QS - 1);
    const int row = threadIdx.x / LN_QS;
    const int nq = ld >> 2;
    const int q = blockIdx.x * LN_QS + ql;
    const int b = blockIdx.y;
    const bool active = q < nq;
    const size_t base = static_cast<size_t>(b) * channels * ld + static_cast<size_t>(q) * 4;

    tile_statistics(x, channels, ld, eps, base, active, row, ql, s_mean, s_m2, s_cnt, s_mu, s_rstd);

    if (threadIdx.x < LN_QS * 4) {
        const int t = blockIdx.x * (LN_QS * 4) + threadIdx.x;
        if (t < ld) {
            const bool live = t < frames;
            float* srow = stats + static_cast<size_t>(b) * 2 * ld;
            srow[t] = live ? s_mu[threadIdx.x] : 0.f;
            srow[ld + t] = live ? s_rstd[threadIdx.x] : 0.f;
        }
    }
}

}  // namespace nbasr

using namespace nbasr;

static int layernorm_impl(const char* what, const float* x, const float* gamma, const float* beta, float* y, float* absmax,
                          int batch, int channels, int frames, int ld, float eps, nbasr_stream_t stream)
{
    clear_error();
    NBASR_REQUIRE(batch >= 0 && channels > 0 && frames >= 0, NBASR_EINVAL, "%s: bad sizes", what);
    NBASR_REQUIRE(ld >= frames && ld % 4 == 0, NBASR_EALIGN, "%s: ld=%d must be >= frames=%d and a multiple of 4", what, ld, frames);
    if (batch == 0 || ld == 0) return NBASR_OK;
    NBASR_REQUIRE(x && gamma && beta && y, NBASR_ENULL, "%s: NULL pointer", what);
    NBASR_REQUIRE(aligned16(x) && aligned16(y), NBASR_EALIGN, "%s: x, y must be 16-byte aligned", what);
    NBASR_REQUIRE(batch <= 65535, NBASR_EINVAL, "%s: batch %d > 65535", what, batch);
    const int nq = ld / 4;
    const dim3 grid((nq + LN_QS - 1) / LN_QS, batch);
    if (absmax) {
        const hipError_t e = hipMemsetAsync(absmax, 0, sizeof(float) * batch, as_stream(stream));
        if (e != hipSuccess) { set_error("%s: hipMemsetAsync failed: %s", what, hipGetErrorString(e)); return static_cast<int>(e); }
        hipLaunchKernelGGL(layernorm_channels_kernel<true>, grid, dim3(256), 0, as_stream(stream),
                           x, gamma, beta, y, channels, frames, ld, eps, reinterpret_cast<unsigned*>(absmax));
    } else {
        hipLaunchKernelGGL(layernorm_channels_kernel<false>, grid, dim3(256), 0, as_stream(stream),
                           x, gamma, beta, y, channels, frames, ld, eps, static_cast<unsigned*>(nullptr));
    }
    return launch_status(what);
}

extern "C" int nbasr_layernorm_channels(const float* x, const float* gamma, const float* beta, float* y, int batch,
                                        int channels, int frames, int ld, float eps, nbasr_stream_t stream)
{
    return layernorm_impl("nbasr_layernorm_channels", x, gamma, beta, y, nullptr, batch, channels, frames, ld, eps, stream);
}

extern "C" int nbasr_layernorm_channels_absmax(const float* x, const float* gamma, const float* beta, float* y, float* absmax,
                                               int batch, int channels, int frames, int ld, float eps, nbasr_stream_t stream)
{
    clear_error();
    NBASR_REQUIRE(absmax != nullptr || batch == 0, NBASR_ENULL, "nbasr_layernorm_channels_absmax: absmax is NULL");
    return layernorm_impl("nbasr_layernorm_channels_absmax", x, gamma, beta, y, absmax, batch, channels, frames, ld, eps, stream);
}

// absmax[b] = max |x[b, :]| over n contiguous floats per utterance (n % 4 == 0): the range information of the 2-way fp16
// dense convolution when its input does not come out of the LayerNorm kernel (the model input)
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, unsigned* __restrict__ absmax, size_t n4)
{
    const int b = blockIdx.y;
    const float4* __restrict__ xb = reinterpret_cast<const float4*>(x) + static_cast<size_t>(b) * n4;
    float m = 0.f;
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n4; i += static_cast<size_t>(gridDim.x) * blockDim.x) {
        const float4 v = xb[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
    if ((threadIdx.x & 63) == 0) atomicMax(absmax + b, __float_as_uint(m));
}

extern "C" int nbasr_absmax(const float* x, float* absmax, int batch, long long n, nbasr_stream_t stream)
{
    clear_error();
    NBASR_REQUIRE(batch >= 0 && n >= 0 && n % 4 == 0, NBASR_EINVAL, "nbasr_absmax: batch >= 0 and n %% 4 == 0 required (n=%lld)", n);
    if (batch == 0) return NBASR_OK;
    NBASR_REQUIRE(absmax, NBASR_ENULL, "nbasr_absmax: absmax is NULL");
    const hipError_t e = hipMemsetAsync(absmax, 0, sizeof(float) * batch, as_stream(stream));
    if (e != hipSuccess) { set_error("nbasr_absmax: hipMemsetAsync failed: %s", hipGetErrorString(e)); return static_cast<int>(e); }
    if (n == 0) return NBASR_OK;
    NBASR_REQUIRE(x, NBASR_ENULL, "nbasr_absmax: x is NULL");
    NBASR_REQUIRE(aligned16(x), NBASR_EALIGN, "nbasr_absmax: x must be 16-byte aligned");
    NBASR_REQUIRE(batch <= 65535, NBASR_EINVAL, "nbasr_absmax: batch %d > 65535", batch);
    const size_t n4 = static_cast<size_t>(n / 4);
    const unsigned gx = static_cast<unsigned>(n4 / 1024 + 1 < 64 ? n4 / 1024 + 1 : 64);
    hipLaunchKernelGGL(absmax_kernel, dim3(gx, batch), dim3(256), 0, as_stream(stream), x, reinterpret_cast<unsigned*>(absmax), n4);
    return launch_status("nbasr_absmax");
}

extern "C" int nbasr_channel_stats(const float* x, float* stats, int batch, int channels, int frames, int ld, float eps,
                                   nbasr_stream_t stream)
{
    clear_error();
    NBASR_REQUIRE(batch >= 0 && channels > 0 && frames >= 0, NBASR_EINVAL, "nbasr_channel_stats: bad sizes");
    NBASR_REQUIRE(ld >= frames && ld % 4 == 0, NBASR_EALIGN, "nbasr_channel_stats: ld=%d must be >= frames=%d and a multiple of 4", ld, frames);
    if (batch == 0 || ld == 0) return NBASR_OK;
    NBASR_REQUIRE(x && stats, NBASR_ENULL, "nbasr_channel_stats: NULL pointer");
    NBASR_REQUIRE(aligned16(x) && aligned16(stats), NBASR_EALIGN, "nbasr_channel_stats: x, stats must be 16-byte aligned");
    NBASR_REQUIRE(batch <= 65535, NBASR_EINVAL, "nbasr_channel_stats: batch %d > 65535", batch);
    const int nq = ld / 4;
    hipLaunchKernelGGL(channel_stats_kernel, dim3((nq + LN_QS - 1) / LN_QS, batch), dim3(256), 0, as_stream(stream),
                       x, stats, channels, frames, ld, eps);
    return launch_status("nbasr_channel_stats");
}
