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
#include "storage.h"

namespace nbasr {

constexpr int LN_ROWS = 16;   // channel rows in flight per workgroup
constexpr int LN_QS = 16;     // 16-byte chunks (4 fp32 / 8 bf16 frames each) per row segment

// Statistics of one 64-frame tile: fills s_mu / s_rstd (LDS) for the tile's frame columns.  Called by all 256 threads.
// MINMAX: also the per-frame extrema of x over the channels (s_lo / s_hi: [LN_ROWS][64] partials in, merged into row 0).
template <typename T, bool MINMAX = false, int FR = Chunk<T>::FR>
__device__ __forceinline__ void tile_statistics(const T* x, int channels, int ld, float eps, size_t base, bool active,
                                                int row, int ql, float (*s_mean)[LN_QS * FR], float (*s_m2)[LN_QS * FR],
                                                float* s_cnt, float* s_mu, float* s_rstd,
                                                float (*s_lo)[LN_QS * FR] = nullptr, float (*s_hi)[LN_QS * FR] = nullptr)
{
    // per-lane shifted sums over this lane's channel subset
    float shift[FR], s1[FR], s2[FR], lo[FR], hi[FR];
#pragma unroll
    for (int r = 0; r < FR; ++r) { shift[r] = 0.f; s1[r] = 0.f; s2[r] = 0.f; lo[r] = 3.0e38f; hi[r] = -3.0e38f; }
    int n = 0;
    if (active && row < channels) {
        // the shift is the lane's first channel (re-read in the loop: an L1 hit); four rows in flight per lane -- with one, a
        // workgroup had 4 KiB outstanding and the kernels that call this ran at 3.6 TB/s
        load_frames<FR>(x + base + static_cast<size_t>(row) * ld, shift);
        auto fold = [&](const float (&e)[FR]) {
#pragma unroll
            for (int r = 0; r < FR; ++r) {
                const float d = e[r] - shift[r];
                s1[r] += d;
                s2[r] = __builtin_fmaf(d, d, s2[r]);
                if (MINMAX) { lo[r] = fminf(lo[r], e[r]); hi[r] = fmaxf(hi[r], e[r]); }
            }
            ++n;
        };
        int c = row;
        // A lane walks channels / 16 rows: with four rows in flight that is 12 dependent round trips at 800 channels -- ~35 us whatever the
        // bytes (bf16 rows: 1.7 TB/s; few workgroups per CU, so nothing else hides them).  Eight in flight halve the trips; the folds stay
        // in channel order, so the sums are the same bit for bit.
        for (; c + 7 * LN_ROWS < channels; c += 8 * LN_ROWS) {
            float e[8][FR];
#pragma unroll
            for (int u = 0; u < 8; ++u) load_frames<FR>(x + base + static_cast<size_t>(c + u * LN_ROWS) * ld, e[u]);
#pragma unroll
            for (int u = 0; u < 8; ++u) fold(e[u]);
        }
        for (; c + 3 * LN_ROWS < channels; c += 4 * LN_ROWS) {       // four loads issued, then four folds (same order as one by one)
            float e[4][FR];
#pragma unroll
            for (int u = 0; u < 4; ++u) load_frames<FR>(x + base + static_cast<size_t>(c + u * LN_ROWS) * ld, e[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u) fold(e[u]);
        }
        for (; c < channels; c += LN_ROWS) {
            float e[FR];
            load_frames<FR>(x + base + static_cast<size_t>(c) * ld, e);
            fold(e);
        }
    }
    const float fn = static_cast<float>(n);
#pragma unroll
    for (int r = 0; r < FR; ++r) {
        const float dm = n ? s1[r] / fn : 0.f;
        s_mean[row][ql * FR + r] = shift[r] + dm;
        s_m2[row][ql * FR + r] = n ? fmaxf(s2[r] - dm * s1[r], 0.f) : 0.f;
        if (MINMAX) { s_lo[row][ql * FR + r] = lo[r]; s_hi[row][ql * FR + r] = hi[r]; }
    }
    if (ql == 0) s_cnt[row] = fn;
    __syncthreads();

    // merge the 16 row partials (Chan's parallel-variance formula), one thread per frame column
    if (threadIdx.x < LN_QS * FR) {
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
        if (MINMAX) {
            float l = s_lo[0][col], h = s_hi[0][col];
#pragma unroll
            for (int k = 1; k < LN_ROWS; ++k) { l = fminf(l, s_lo[k][col]); h = fmaxf(h, s_hi[k][col]); }
            s_lo[0][col] = l; s_hi[0][col] = h;       // only this thread touches column `col` of rows 0..15 here
        }
    }
    __syncthreads();
}

// ABSMAX: also fold max|y| of each utterance into absmax[b] (float bits, non-negative floats order like unsigned ints);
// the 2-way fp16 dense conv (gemm_conv_split.hip) scales its input by a power of two derived from it
// TI / TO: storage types of x and y (float | bf16_t; bf16 -> float feeds the fp32 LSTM projection of the bf16 path)
template <typename TI, typename TO, bool ABSMAX>
__global__ __launch_bounds__(256) void layernorm_channels_kernel(
    const TI* x, const float* __restrict__ gamma, const float* __restrict__ beta,
    TO* y, int channels, int frames, int ld, float eps, unsigned* __restrict__ absmax)   // x may alias y (in-place, TI == TO)
{
    constexpr int FR = Chunk<TI>::FR;
    __shared__ float s_mean[LN_ROWS][LN_QS * FR];
    __shared__ float s_m2[LN_ROWS][LN_QS * FR];
    __shared__ float s_cnt[LN_ROWS];
    __shared__ float s_mu[LN_QS * FR];
    __shared__ float s_rstd[LN_QS * FR];

    const int ql = threadIdx.x & (LN_QS - 1);
    const int row = threadIdx.x / LN_QS;
    const int nq = ld / FR;
    const int q = blockIdx.x * LN_QS + ql;
    const int b = blockIdx.y;
    const bool active = q < nq;
    const size_t base = static_cast<size_t>(b) * channels * ld + static_cast<size_t>(q) * FR;

    tile_statistics<TI>(x, channels, ld, eps, base, active, row, ql, s_mean, s_m2, s_cnt, s_mu, s_rstd);

    // pass 2: normalise, scale, shift; keep the pitch columns at zero
    if (!active && !ABSMAX) return;
    float mu[FR], rs[FR];
#pragma unroll
    for (int r = 0; r < FR; ++r) { mu[r] = s_mu[ql * FR + r]; rs[r] = s_rstd[ql * FR + r]; }
    const int t0 = q * FR;
    float amax = 0.f;
    for (int c = row; active && c < channels; c += LN_ROWS) {
        const size_t off = base + static_cast<size_t>(c) * ld;
        float o[FR];
        load_frames<FR>(x + off, o);
        const float g = gamma[c], bt = beta[c];
#pragma unroll
        for (int r = 0; r < FR; ++r) {
            o[r] = (o[r] - mu[r]) * rs[r] * g + bt;
            if (t0 + r >= frames) o[r] = 0.f;
            if (ABSMAX) amax = fmaxf(amax, finite_abs(o[r]));
        }
        store_frames<FR, false>(y + off, o);
    }
    if (ABSMAX) {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) amax = fmaxf(amax, __shfl_xor(amax, m));
        if ((threadIdx.x & 63) == 0) atomicMax(absmax + b, __float_as_uint(amax));
    }
}

// deferred LayerNorm: ONE read pass, per-frame (mean, rstd) only; consumers normalise while loading
template <typename T>
__global__ __launch_bounds__(256) void channel_stats_kernel(const T* __restrict__ x, float* __restrict__ stats,
                                                            int channels, int frames, int ld, float eps)
{
    constexpr int FR = Chunk<T>::FR;
    __shared__ float s_mean[LN_ROWS][LN_QS * FR];
    __shared__ float s_m2[LN_ROWS][LN_QS * FR];
    __shared__ float s_cnt[LN_ROWS];
    __shared__ float s_mu[LN_QS * FR];
    __shared__ float s_rstd[LN_QS * FR];

    const int ql = threadIdx.x & (LN_QS - 1);
    const int row = threadIdx.x / LN_QS;
    const int nq = ld / FR;
    const int q = blockIdx.x * LN_QS + ql;
    const int b = blockIdx.y;
    const bool active = q < nq;
    const size_t base = static_cast<size_t>(b) * channels * ld + static_cast<size_t>(q) * FR;

    tile_statistics<T>(x, channels, ld, eps, base, active, row, ql, s_mean, s_m2, s_cnt, s_mu, s_rstd);

    if (threadIdx.x < LN_QS * FR) {
        const int t = blockIdx.x * (LN_QS * FR) + threadIdx.x;
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

static int layernorm_impl(const char* what, const void* x, const float* gamma, const float* beta, void* y, float* absmax,
                          int batch, int channels, int frames, int ld, float eps, nbasr_stream_t stream,
                          int in_dtype = NBASR_F32, int out_dtype = NBASR_F32)
{
    clear_error();
    NBASR_REQUIRE((in_dtype == NBASR_F32 || in_dtype == NBASR_BF16) && (out_dtype == NBASR_F32 || out_dtype == NBASR_BF16) &&
                  !(in_dtype == NBASR_F32 && out_dtype == NBASR_BF16), NBASR_EINVAL,
                  "%s: storage types (%d -> %d) unsupported (f32 -> f32, bf16 -> bf16, bf16 -> f32)", what, in_dtype, out_dtype);
    NBASR_REQUIRE(batch >= 0 && channels > 0 && frames >= 0, NBASR_EINVAL, "%s: bad sizes", what);
    const int fr = in_dtype == NBASR_BF16 ? 8 : 4;
    NBASR_REQUIRE(ld >= frames && ld % fr == 0, NBASR_EALIGN, "%s: ld=%d must be >= frames=%d and a multiple of %d", what, ld, frames, fr);
    if (batch == 0 || ld == 0) return NBASR_OK;
    NBASR_REQUIRE(x && gamma && beta && y, NBASR_ENULL, "%s: NULL pointer", what);
    NBASR_REQUIRE(aligned16(x) && aligned16(y), NBASR_EALIGN, "%s: x, y must be 16-byte aligned", what);
    NBASR_REQUIRE(batch <= 65535, NBASR_EINVAL, "%s: batch %d > 65535", what, batch);
    NBASR_REQUIRE(!absmax || in_dtype == NBASR_F32, NBASR_EINVAL, "%s: the absmax by-product exists for fp32 tensors only", what);
    const int nq = ld / fr;
    const dim3 grid((nq + LN_QS - 1) / LN_QS, batch);
    if (in_dtype == NBASR_BF16) {
        if (out_dtype == NBASR_BF16)
            hipLaunchKernelGGL((layernorm_channels_kernel<bf16_t, bf16_t, false>), grid, dim3(256), 0, as_stream(stream),
                               static_cast<const bf16_t*>(x), gamma, beta, static_cast<bf16_t*>(y), channels, frames, ld, eps, static_cast<unsigned*>(nullptr));
        else
            hipLaunchKernelGGL((layernorm_channels_kernel<bf16_t, float, false>), grid, dim3(256), 0, as_stream(stream),
                               static_cast<const bf16_t*>(x), gamma, beta, static_cast<float*>(y), channels, frames, ld, eps, static_cast<unsigned*>(nullptr));
    } else if (absmax) {
        zero_async(absmax, sizeof(float) * batch, as_stream(stream));
        hipLaunchKernelGGL((layernorm_channels_kernel<float, float, true>), grid, dim3(256), 0, as_stream(stream),
                           static_cast<const float*>(x), gamma, beta, static_cast<float*>(y), channels, frames, ld, eps, reinterpret_cast<unsigned*>(absmax));
    } else {
        hipLaunchKernelGGL((layernorm_channels_kernel<float, float, false>), grid, dim3(256), 0, as_stream(stream),
                           static_cast<const float*>(x), gamma, beta, static_cast<float*>(y), channels, frames, ld, eps, static_cast<unsigned*>(nullptr));
    }
    return launch_status(what);
}

extern "C" int nbasr_layernorm_channels(const void* x, const float* gamma, const float* beta, void* y, float* absmax, int batch,
                                        int channels, int frames, int ld, float eps, int in_dtype, int out_dtype, nbasr_stream_t stream)
{
    clear_error();
    NBASR_REQUIRE(!absmax || (in_dtype == NBASR_F32 && out_dtype == NBASR_F32), NBASR_EINVAL,
                  "nbasr_layernorm_channels: the max|y| by-product exists for fp32 -> fp32 only");
    return layernorm_impl("nbasr_layernorm_channels", x, gamma, beta, y, absmax, batch, channels, frames, ld, eps, stream, in_dtype, out_dtype);
}

// ---- LayerNorm whose consumer is the fp16-split dense convolution: statistics + range bound, then normalise + split ------
// The convolution wants its input as a pre-split fp16 image scaled by ONE power of two per utterance (gemm_conv_split.hip,
// XIMG).  The scale must be known before the first row is written, so the LayerNorm runs as two kernels with the same total
// traffic as the fused one (2 reads, the second L2-warm, + 1 write):
//   1. channel_stats_bound_kernel: per-frame (mean, rstd) as channel_stats_kernel, and an upper bound of the normalised
//      magnitudes  max_c |x_c - mean| * rstd * max|gamma| + max|beta|  (from the per-frame extrema tracked in the same
//      pass) folded into bound[b] (atomicMax on float bits);
//   2. normalize_split_kernel: y = (x - mean) rstd gamma + beta, scaled by 2^k(bound[b]), split v = hi + lo (two fp16 terms, lo unscaled), written
//      as image[b][16-channel group][split][8-channel half][1 + ld rows][8 ch] (row 0 zero, frame t at row t + 1).
__global__ __launch_bounds__(256) void channel_stats_bound_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta, float* __restrict__ stats,
                                                                  unsigned* __restrict__ bound, int channels, int frames, int ld, float eps)
{
    __shared__ float s_mean[LN_ROWS][LN_QS * 4];
    __shared__ float s_m2[LN_ROWS][LN_QS * 4];
    __shared__ float s_cnt[LN_ROWS];
    __shared__ float s_mu[LN_QS * 4];
    __shared__ float s_rstd[LN_QS * 4];
    __shared__ float s_gb[8];
    __shared__ float s_lo[LN_ROWS][LN_QS * 4];
    __shared__ float s_hi[LN_ROWS][LN_QS * 4];

    const int ql = threadIdx.x & (LN_QS - 1);
    const int row = threadIdx.x / LN_QS;
    const int nq = ld >> 2;
    const int q = blockIdx.x * LN_QS + ql;
    const int b = blockIdx.y;
    const bool active = q < nq;
    const size_t base = static_cast<size_t>(b) * channels * ld + static_cast<size_t>(q) * 4;

    // max|gamma|, max|beta| (every workgroup recomputes them: `channels` values)
    float gm = 0.f, bm = 0.f;
    for (int c = threadIdx.x; c < channels; c += 256) { gm = fmaxf(gm, fabsf(gamma[c])); bm = fmaxf(bm, fabsf(beta[c])); }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { gm = fmaxf(gm, __shfl_xor(gm, d)); bm = fmaxf(bm, __shfl_xor(bm, d)); }
    if ((threadIdx.x & 63) == 0) { s_gb[threadIdx.x >> 6] = gm; s_gb[4 + (threadIdx.x >> 6)] = bm; }

    tile_statistics<float, true>(x, channels, ld, eps, base, active, row, ql, s_mean, s_m2, s_cnt, s_mu, s_rstd, s_lo, s_hi);   // ends with a barrier
    gm = fmaxf(fmaxf(s_gb[0], s_gb[1]), fmaxf(s_gb[2], s_gb[3]));
    bm = fmaxf(fmaxf(s_gb[4], s_gb[5]), fmaxf(s_gb[6], s_gb[7]));

    float dev = 0.f;
    if (threadIdx.x < LN_QS * 4) {
        const int t = blockIdx.x * (LN_QS * 4) + threadIdx.x;
        if (t < ld) {
            const bool live = t < frames;
            float* srow = stats + static_cast<size_t>(b) * 2 * ld;
            srow[t] = live ? s_mu[threadIdx.x] : 0.f;
            srow[ld + t] = live ? s_rstd[threadIdx.x] : 0.f;
            // max_c |x_c - mean| = max(hi - mean, mean - lo): no second pass over the tile
            if (live) dev = finite_abs(fmaxf(s_hi[0][threadIdx.x] - s_mu[threadIdx.x], s_mu[threadIdx.x] - s_lo[0][threadIdx.x]) * s_rstd[threadIdx.x]);
        }
    }
    if (threadIdx.x < 64) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) dev = fmaxf(dev, __shfl_xor(dev, d));
        // a little headroom for the rounding of the bound itself and of the normalisation it bounds
        if (threadIdx.x == 0) atomicMax(bound + b, __float_as_uint(finite_abs((dev * gm + bm) * 1.0001f)));
    }
}

// NORM = false: no LayerNorm, x itself is scaled and split (the MODEL INPUT in front of conv 0; bound = the per-utterance maximum
// that nbasr_input_range wrote, one every `bound_stride` floats).
template <bool NORM>
__global__ __launch_bounds__(256) void normalize_split_kernel(const float* __restrict__ x, const float* __restrict__ stats,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              const float* __restrict__ bound, unsigned char* __restrict__ image,
                                                              int channels, int /* frames: rstd is 0 beyond them */, int ld,
                                                              int bound_stride)
{
    typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));
    // 64 frame quads (one wave = 1 KiB of a channel row) x 4 channel octets per pass, the octets dealt over the waves of the
    // gridDim.z workgroups of a tile; a thread turns 8 channels x 4 frames into 4 + 4 image rows of 16 bytes.  (Round 1 had 16 quads
    // x 16 octets: 256-byte row segments, 3.7 TB/s.)
    const int ql = threadIdx.x & 63, oct0 = (threadIdx.x >> 6) + 4 * blockIdx.z;
    const int nq = ld >> 2;
    const int q = blockIdx.x * 64 + ql;
    const int b = blockIdx.y;
    const int n_groups = (channels + 15) >> 4;
    const size_t rows = static_cast<size_t>(ld) + 1;
    unsigned char* const img_b = image + static_cast<size_t>(b) * n_groups * 4 * rows * 16;
    // the zero rows (row 0 of every plane): one workgroup per utterance writes them
    if (blockIdx.x == 0 && blockIdx.z == 0) {
        const uint4 z = make_uint4(0u, 0u, 0u, 0u);
        for (int pl = threadIdx.x; pl < n_groups * 4; pl += 256) *reinterpret_cast<uint4*>(img_b + static_cast<size_t>(pl) * rows * 16) = z;
    }
    if (q >= nq) return;
    // 2^k that brings the bound into [2^14, 2^15)
    float scale = 1.f;
    {
        const unsigned bits = __float_as_uint(bound[static_cast<size_t>(b) * bound_stride]);
        const int e = static_cast<int>((bits >> 23) & 0xffu);
        int k = (bits & 0x7fffffffu) ? (127 + 14) - e : 0;
        k = k > 126 ? 126 : (k < -126 ? -126 : k);
        scale = __uint_as_float(static_cast<unsigned>(127 + k) << 23);
    }
    float m4[4] = {0.f, 0.f, 0.f, 0.f}, r4[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (NORM) {
        const float* st = stats + static_cast<size_t>(b) * 2 * ld + q * 4;
        const float4 mu = *reinterpret_cast<const float4*>(st), rs = *reinterpret_cast<const float4*>(st + ld);
        m4[0] = mu.x; m4[1] = mu.y; m4[2] = mu.z; m4[3] = mu.w;
        r4[0] = rs.x; r4[1] = rs.y; r4[2] = rs.z; r4[3] = rs.w;
    }
    const float* xb = x + static_cast<size_t>(b) * channels * ld + static_cast<size_t>(q) * 4;
    const int n_oct = (channels + 7) >> 3;
    for (int oct = oct0; oct < 2 * n_groups; oct += 4 * gridDim.z) {
        float v[8][4];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int ch = oct * 8 + c;
            float4 t4 = make_float4(0.f, 0.f, 0.f, 0.f);
            float g = 0.f, be = 0.f;
            if (oct < n_oct && ch < channels) {
                t4 = *reinterpret_cast<const float4*>(xb + static_cast<size_t>(ch) * ld);
                if constexpr (NORM) { g = gamma[ch]; be = beta[ch]; }
            }
            const float e4[4] = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if constexpr (NORM) v[c][r] = (ch < channels) ? ln_apply(e4[r], m4[r], r4[r], g, be) * scale : 0.f;
                else                v[c][r] = e4[r] * scale;
            }
        }
        // group = oct >> 1, half = oct & 1; planes of a group: [split][half]
        unsigned char* plane_hi = img_b + ((static_cast<size_t>(oct >> 1) * 4 + (oct & 1)) * rows + 1 + static_cast<size_t>(q) * 4) * 16;
        unsigned char* plane_lo = plane_hi + 2 * rows * 16;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            halfx8 hi, lo;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const _Float16 h = static_cast<_Float16>(v[c][r]);
                hi[c] = h;
                lo[c] = static_cast<_Float16>(v[c][r] - static_cast<float>(h));       // unscaled residual (gemm_conv_split.hip, SplitF16x2)
            }
            *reinterpret_cast<halfx8*>(plane_hi + r * 16) = hi;
            *reinterpret_cast<halfx8*>(plane_lo + r * 16) = lo;
        }
    }
}

// workgroups that share one (64-quad tile, utterance) of normalize_split_kernel, each taking every z-th set of 4 channel octets: 4 at the
// benchmark batch (1 024 workgroups); up to 16 where a small batch would otherwise leave most compute units without one (8 utterances x
// 1000 frames: 128 workgroups -> 512; the same bytes in half the time, the kernel is three launches of a forward)
static int split_grid_z(int tiles, int batch)
{
    const long long wgs = static_cast<long long>(tiles) * batch;
    int z = 4;
    while (z < 16 && wgs * z < 512) z *= 2;
    return z;
}

extern "C" int nbasr_layernorm_split_image(const float* x, const float* gamma, const float* beta, float* stats, float* bound,
                                           void* image, int batch, int channels, int frames, int ld, float eps,
                                           nbasr_stream_t stream)
{
    clear_error();
    NBASR_REQUIRE(batch >= 0 && channels > 0 && frames >= 0, NBASR_EINVAL, "nbasr_layernorm_split_image: bad sizes");
    NBASR_REQUIRE(ld >= frames && ld % 4 == 0, NBASR_EALIGN, "nbasr_layernorm_split_image: ld=%d must be >= frames=%d and a multiple of 4", ld, frames);
    if (batch == 0 || ld == 0) return NBASR_OK;
    NBASR_REQUIRE(x && gamma && beta && stats && bound && image, NBASR_ENULL, "nbasr_layernorm_split_image: NULL pointer");
    NBASR_REQUIRE(aligned16(x) && aligned16(stats) && aligned16(image), NBASR_EALIGN, "nbasr_layernorm_split_image: x, stats, image must be 16-byte aligned");
    NBASR_REQUIRE(batch <= 65535, NBASR_EINVAL, "nbasr_layernorm_split_image: batch %d > 65535", batch);
    zero_async(bound, sizeof(float) * batch, as_stream(stream));
    const int nq = ld / 4;
    hipLaunchKernelGGL(channel_stats_bound_kernel, dim3((nq + LN_QS - 1) / LN_QS, batch), dim3(256), 0, as_stream(stream),
                       x, gamma, beta, stats, reinterpret_cast<unsigned*>(bound), channels, frames, ld, eps);
    hipLaunchKernelGGL(normalize_split_kernel<true>, dim3((nq + 63) / 64, batch, split_grid_z((nq + 63) / 64, batch)), dim3(256), 0, as_stream(stream),
                       x, stats, gamma, beta, bound, static_cast<unsigned char*>(image), channels, frames, ld, 1);
    return launch_status("nbasr_layernorm_split_image");
}

// Range summary of a caller-supplied tensor (the MODEL INPUT, the one activation whose range this library does not control):
// range[b] = { max finite |x[b]|,  min over frames of (max over channels of |x[b, :, t]|) among frames where that is > 0,
//              non-zero if x[b] holds an Inf or NaN,  unused }.
// The scaled fp16 convolution keeps full fp32 precision for elements down to 2^-29 of the utterance's maximum; an utterance
// whose quietest frame lies more than 2^20 below its loudest sample (or that holds non-finite values) is EXTREME and is
// routed to the range-free 3-way bf16 split instead (nbasr_dense_conv1d_packed with x_range), per utterance, on the device.
// 64 lanes x 4 frames wide, 4 channel slices deep: 16-byte loads, all of a thread's loads independent
__global__ __launch_bounds__(256) void input_range_kernel(const float* __restrict__ x, unsigned* __restrict__ range,
                                                          int channels, int frames, int ld)
{
    __shared__ float s_max[4][256];
    __shared__ unsigned s_bad[4];
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;             // slice: wave-uniform
    const int t0 = (blockIdx.x * 64 + lane) * 4;
    float m[4] = {0.f, 0.f, 0.f, 0.f};
    unsigned bad = 0u;
    if (t0 < frames) {
        const float* col = x + static_cast<size_t>(b) * channels * ld + t0;
        const bool whole = t0 + 4 <= ld && ld % 4 == 0 && (reinterpret_cast<size_t>(x) & 15) == 0;
        for (int c = slice; c < channels; c += 4) {
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (whole) {
                const float4 q = *reinterpret_cast<const float4*>(col + static_cast<size_t>(c) * ld);
                v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (t0 + r < frames) v[r] = col[static_cast<size_t>(c) * ld + r];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float a = fabsf(v[r]);
                if (t0 + r < frames) {
                    if (!(a <= 3.4028234664e38f)) bad = 1u;      // Inf or NaN
                    else m[r] = fmaxf(m[r], a);
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) s_max[slice][lane * 4 + r] = m[r];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) bad |= __shfl_xor(bad, d);
    if (lane == 0) s_bad[slice] = bad;
    __syncthreads();
    // one frame per thread: its maximum over all channels, then the tile's loudest sample and quietest non-silent frame
    const int t = blockIdx.x * 256 + threadIdx.x;
    const float fmax_ = fmaxf(fmaxf(s_max[0][threadIdx.x], s_max[1][threadIdx.x]), fmaxf(s_max[2][threadIdx.x], s_max[3][threadIdx.x]));
    float hi = fmax_, lo = (t < frames && fmax_ > 0.f) ? fmax_ : __uint_as_float(0x7f800000u);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        hi = fmaxf(hi, __shfl_xor(hi, d));
        lo = fminf(lo, __shfl_xor(lo, d));
    }
    if (lane == 0) {
        atomicMax(range + 4 * b, __float_as_uint(hi));        // non-negative floats order like unsigned integers
        atomicMin(range + 4 * b + 1, __float_as_uint(lo));
        if (slice == 0 && (s_bad[0] | s_bad[1] | s_bad[2] | s_bad[3])) atomicOr(range + 4 * b + 2, 0x3f800000u);     // 1.0f
    }
}

__global__ void input_range_init_kernel(unsigned* __restrict__ range, int batch)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < batch) { range[4 * b] = 0u; range[4 * b + 1] = 0x7f800000u; range[4 * b + 2] = 0u; range[4 * b + 3] = 0u; }
}

extern "C" int nbasr_input_range(const float* x, float* range, int batch, int channels, int frames, int ld, nbasr_stream_t stream)
{
    clear_error();
    NBASR_REQUIRE(batch >= 0 && channels > 0 && frames >= 0 && ld >= frames, NBASR_EINVAL, "nbasr_input_range: bad sizes");
    if (batch == 0) return NBASR_OK;
    NBASR_REQUIRE(range, NBASR_ENULL, "nbasr_input_range: range is NULL");
    NBASR_REQUIRE(batch <= 65535, NBASR_EINVAL, "nbasr_input_range: batch %d > 65535", batch);
    hipLaunchKernelGGL(input_range_init_kernel, dim3((batch + 255) / 256), dim3(256), 0, as_stream(stream), reinterpret_cast<unsigned*>(range), batch);
    if (frames > 0) {
        NBASR_REQUIRE(x, NBASR_ENULL, "nbasr_input_range: x is NULL");
        hipLaunchKernelGGL(input_range_kernel, dim3((frames + 255) / 256, batch), dim3(256), 0, as_stream(stream), x,
                           reinterpret_cast<unsigned*>(range), channels, frames, ld);
    }
    return launch_status("nbasr_input_range");
}

extern "C" int nbasr_split_image_ranged(const float* x, const float* x_range, void* image, int batch, int channels, int frames,
                                        int ld, nbasr_stream_t stream)
{
    clear_error();
    NBASR_REQUIRE(batch >= 0 && channels > 0 && frames >= 0, NBASR_EINVAL, "nbasr_split_image_ranged: bad sizes");
    NBASR_REQUIRE(ld >= frames && ld % 4 == 0, NBASR_EALIGN, "nbasr_split_image_ranged: ld=%d must be >= frames=%d and a multiple of 4", ld, frames);
    if (batch == 0 || ld == 0) return NBASR_OK;
    NBASR_REQUIRE(x && x_range && image, NBASR_ENULL, "nbasr_split_image_ranged: NULL pointer");
    NBASR_REQUIRE(aligned16(x) && aligned16(image), NBASR_EALIGN, "nbasr_split_image_ranged: x, image must be 16-byte aligned");
    NBASR_REQUIRE(batch <= 65535, NBASR_EINVAL, "nbasr_split_image_ranged: batch %d > 65535", batch);
    const int nq = ld / 4;
    hipLaunchKernelGGL(normalize_split_kernel<false>, dim3((nq + 63) / 64, batch, split_grid_z((nq + 63) / 64, batch)), dim3(256), 0, as_stream(stream),
                       x, static_cast<const float*>(nullptr), static_cast<const float*>(nullptr), static_cast<const float*>(nullptr),
                       x_range, static_cast<unsigned char*>(image), channels, frames, ld, 4);
    return launch_status("nbasr_split_image_ranged");
}

extern "C" int nbasr_channel_stats(const void* x, float* stats, int batch, int channels, int frames, int ld, float eps, int dtype,
                                     nbasr_stream_t stream)
{
    clear_error();
    NBASR_REQUIRE(dtype == NBASR_F32 || dtype == NBASR_BF16, NBASR_EINVAL, "nbasr_channel_stats: unknown dtype %d", dtype);
    NBASR_REQUIRE(batch >= 0 && channels > 0 && frames >= 0, NBASR_EINVAL, "nbasr_channel_stats: bad sizes");
    const int fr = dtype == NBASR_BF16 ? 8 : 4;
    NBASR_REQUIRE(ld >= frames && ld % fr == 0, NBASR_EALIGN, "nbasr_channel_stats: ld=%d must be >= frames=%d and a multiple of %d", ld, frames, fr);
    if (batch == 0 || ld == 0) return NBASR_OK;
    NBASR_REQUIRE(x && stats, NBASR_ENULL, "nbasr_channel_stats: NULL pointer");
    NBASR_REQUIRE(aligned16(x) && aligned16(stats), NBASR_EALIGN, "nbasr_channel_stats: x, stats must be 16-byte aligned");
    NBASR_REQUIRE(batch <= 65535, NBASR_EINVAL, "nbasr_channel_stats: batch %d > 65535", batch);
    const int nq = ld / fr;
    const dim3 grid((nq + LN_QS - 1) / LN_QS, batch);
    if (dtype == NBASR_BF16)
        hipLaunchKernelGGL(channel_stats_kernel<bf16_t>, grid, dim3(256), 0, as_stream(stream), static_cast<const bf16_t*>(x), stats, channels, frames, ld, eps);
    else
        hipLaunchKernelGGL(channel_stats_kernel<float>, grid, dim3(256), 0, as_stream(stream), static_cast<const float*>(x), stats, channels, frames, ld, eps);
    return launch_status("nbasr_channel_stats");
}

// ---- bf16 path: the dense convolution's operand image in ONE bf16 term --------------------------------------------------
// image[b][16-channel group][8-channel half][1 + ld rows][8 ch] bfloat16 (row 0 zero, frame t at row t + 1): the layout the
// fp16-split image has per term, so gemm_conv_split.hip's image kernel gathers it by LDS-DMA in exactly the same way.
// NORM: y = LayerNorm(x) from precomputed per-frame statistics (the LayerNorm in front of downsample convs 1-3);
// otherwise a plain re-layout (the model input in front of conv 0; cells without LayerNorm).
template <typename T, bool NORM>
__global__ __launch_bounds__(256) void bf16_image_kernel(const T* __restrict__ x, const float* __restrict__ stats,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         unsigned char* __restrict__ image, int channels, int frames, int ld)
{
    constexpr int FR = Chunk<T>::FR;
    // 16 frame chunks x 16 channel octets per pass; a thread turns 8 channels x FR frames into FR image rows of 16 bytes
    const int ql = threadIdx.x & 15, oct0 = threadIdx.x >> 4;
    const int nq = ld / FR;
    const int q = blockIdx.x * 16 + ql;
    const int b = blockIdx.y;
    const int n_groups = (channels + 15) >> 4;
    const size_t rows = static_cast<size_t>(ld) + 1;
    unsigned char* const img_b = image + static_cast<size_t>(b) * n_groups * 2 * rows * 16;
    if (blockIdx.x == 0) {            // the zero rows (row 0 of every plane): one workgroup per utterance writes them
        const uint4 z = make_uint4(0u, 0u, 0u, 0u);
        for (int pl = threadIdx.x; pl < n_groups * 2; pl += 256) *reinterpret_cast<uint4*>(img_b + static_cast<size_t>(pl) * rows * 16) = z;
    }
    if (q >= nq) return;
    float m[FR], r[FR];
#pragma unroll
    for (int e = 0; e < FR; ++e) { m[e] = 0.f; r[e] = 1.f; }
    if (NORM) {
        const float* st = stats + static_cast<size_t>(b) * 2 * ld + q * FR;
        load_frames<FR>(st, m);
        load_frames<FR>(st + ld, r);
    }
    const T* xb = x + static_cast<size_t>(b) * channels * ld + static_cast<size_t>(q) * FR;
    const int n_oct = (channels + 7) >> 3;
    for (int oct = oct0; oct < 2 * n_groups; oct += 16) {
        float v[8][FR];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int ch = oct * 8 + c;
#pragma unroll
            for (int e = 0; e < FR; ++e) v[c][e] = 0.f;
            if (oct < n_oct && ch < channels) {
                load_frames<FR>(xb + static_cast<size_t>(ch) * ld, v[c]);
                if (NORM) {
                    const float g = gamma[ch], be = beta[ch];
#pragma unroll
                    for (int e = 0; e < FR; ++e) v[c][e] = ln_apply(v[c][e], m[e], r[e], g, be);
                } else {
#pragma unroll
                    for (int e = 0; e < FR; ++e) if (q * FR + e >= frames) v[c][e] = 0.f;     // pitch columns of a caller's tensor
                }
            }
        }
        unsigned char* plane = img_b + ((static_cast<size_t>(oct >> 1) * 2 + (oct & 1)) * rows + 1 + static_cast<size_t>(q) * FR) * 16;
#pragma unroll
        for (int e = 0; e < FR; ++e) {
            const u4v row = {pack_bf16x2(v[0][e], v[1][e]), pack_bf16x2(v[2][e], v[3][e]), pack_bf16x2(v[4][e], v[5][e]), pack_bf16x2(v[6][e], v[7][e])};
            *reinterpret_cast<u4v*>(plane + e * 16) = row;
        }
    }
}

extern "C" size_t nbasr_bf16_image_bytes(int batch, int channels, int ld)
{
    if (batch <= 0 || channels <= 0 || ld < 0) return 0;
    return static_cast<size_t>(batch) * ((channels + 15) / 16) * 2 * (static_cast<size_t>(ld) + 1) * 16;
}

// stats != NULL: LayerNorm(x) -> image (statistics pass + normalise-and-pack pass; stats (batch, 2, ld) is filled on the way);
// gamma == NULL: plain re-layout of x
extern "C" int nbasr_bf16_image(const void* x, const float* gamma, const float* beta, float* stats, void* image, int batch,
                                int channels, int frames, int ld, float eps, int dtype, nbasr_stream_t stream)
{
    clear_error();
    NBASR_REQUIRE(dtype == NBASR_F32 || dtype == NBASR_BF16, NBASR_EINVAL, "nbasr_bf16_image: unknown dtype %d", dtype);
    NBASR_REQUIRE(batch >= 0 && channels > 0 && frames >= 0, NBASR_EINVAL, "nbasr_bf16_image: bad sizes");
    const int fr = dtype == NBASR_BF16 ? 8 : 4;
    NBASR_REQUIRE(ld >= frames && ld % fr == 0, NBASR_EALIGN, "nbasr_bf16_image: ld=%d must be >= frames=%d and a multiple of %d", ld, frames, fr);
    if (batch == 0 || ld == 0) return NBASR_OK;
    NBASR_REQUIRE(x && image, NBASR_ENULL, "nbasr_bf16_image: NULL pointer");
    const bool norm = gamma != nullptr;
    NBASR_REQUIRE(!norm || (beta && stats), NBASR_ENULL, "nbasr_bf16_image: LayerNorm needs gamma, beta and the stats buffer");
    NBASR_REQUIRE(aligned16(x) && aligned16(stats) && aligned16(image), NBASR_EALIGN, "nbasr_bf16_image: x, stats, image must be 16-byte aligned");
    NBASR_REQUIRE(batch <= 65535, NBASR_EINVAL, "nbasr_bf16_image: batch %d > 65535", batch);
    const int nq = ld / fr;
    if (norm) {
        const int rc = nbasr_channel_stats(x, stats, batch, channels, frames, ld, eps, dtype, stream);
        if (rc != NBASR_OK) return rc;
    }
    const dim3 grid((nq + 15) / 16, batch);
    unsigned char* img = static_cast<unsigned char*>(image);
#define NBASR_IMG(T, NORM) hipLaunchKernelGGL((bf16_image_kernel<T, NORM>), grid, dim3(256), 0, as_stream(stream), static_cast<const T*>(x), \
                                              stats, gamma, beta, img, channels, frames, ld)
    if (dtype == NBASR_BF16) { if (norm) NBASR_IMG(bf16_t, true); else NBASR_IMG(bf16_t, false); }
    else                     { if (norm) NBASR_IMG(float, true); else NBASR_IMG(float, false); }
#undef NBASR_IMG
    return launch_status("nbasr_bf16_image");
}

// element-wise storage conversion of a pitched (rows, ld) tensor (ld % 8 == 0): the bridge between the bf16 encoder and the
// operators that exist in fp32 only
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void convert_kernel(const TI* __restrict__ x, TO* __restrict__ y, size_t n8)
{
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n8; i += static_cast<size_t>(gridDim.x) * blockDim.x) {
        float v[8];
        load_frames<8>(x + i * 8, v);
        store_frames<8, false>(y + i * 8, v);
    }
}

extern "C" int nbasr_convert(const void* x, void* y, long long n, int in_dtype, int out_dtype, nbasr_stream_t stream)
{
    clear_error();
    NBASR_REQUIRE(n >= 0 && n % 8 == 0, NBASR_EINVAL, "nbasr_convert: n=%lld must be a non-negative multiple of 8", n);
    NBASR_REQUIRE((in_dtype == NBASR_F32 && out_dtype == NBASR_BF16) || (in_dtype == NBASR_BF16 && out_dtype == NBASR_F32), NBASR_EINVAL,
                  "nbasr_convert: conversions are f32 -> bf16 and bf16 -> f32");
    if (n == 0) return NBASR_OK;
    NBASR_REQUIRE(x && y, NBASR_ENULL, "nbasr_convert: NULL pointer");
    NBASR_REQUIRE(aligned16(x) && aligned16(y), NBASR_EALIGN, "nbasr_convert: pointers must be 16-byte aligned");
    const size_t n8 = static_cast<size_t>(n / 8);
    const unsigned blocks = static_cast<unsigned>(n8 / 256 + 1 < 8192 ? n8 / 256 + 1 : 8192);
    if (in_dtype == NBASR_F32)
        hipLaunchKernelGGL((convert_kernel<float, bf16_t>), dim3(blocks), dim3(256), 0, as_stream(stream), static_cast<const float*>(x), static_cast<bf16_t*>(y), n8);
    else
        hipLaunchKernelGGL((convert_kernel<bf16_t, float>), dim3(blocks), dim3(256), 0, as_stream(stream), static_cast<const bf16_t*>(x), static_cast<float*>(y), n8);
    return launch_status("nbasr_convert");
}
