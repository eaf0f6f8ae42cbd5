// Backward building blocks of the encoder's two dominant operators (bottom-up: the node op and the
// LayerNorm first; dense convs, `linear`, LSTM and the model-level autograd are not built yet).
//
//  * grouped PadConvRelu node op (reference ops.py:24-30, groups = 100):  z = min(relu(conv(x) + b), 20)
//        gm = dz * [0 < z < 20]                       (relu / clamp_max_ masks, from the op's OUTPUT: no pre-activation is kept)
//        dx[ci][t]      = sum_{co,j} w[co][ci][j] * gm[co][t + lpad - j*d]                           grouped_dgrad_kernel (vector ALU)
//        dw[co][ci][j]  = sum_{b,t}  gm[b][co][t] * x[b][ci][t - lpad + j*d],   db[co] = sum gm      grouped_wgrad_kernel (fp32 MFMA)
//    The weight gradient is a GEMM whose reduction dimension is (batch, frames): per (group, utterance) a wave multiplies the
//    16 x frames matrix of masked output gradients by the frames x (ci, tap) im2col of x on v_mfma_f32_16x16x4_f32 (exact fp32),
//    one extra all-ones column yields the bias gradient; per-utterance partials are summed by a second kernel in a FIXED order
//    (no atomics: the gradients are bit-reproducible).
//  * LayerNorm over channels (model.py:46-47, 55-58, 92):  y = (x - mu) * rstd * gamma + beta
//        dx = rstd * (g - mean_c(g) - xhat * mean_c(g * xhat)),  g = dy * gamma;   dgamma = sum_{b,t} dy * xhat,  dbeta = sum dy
#include "storage.h"

namespace nbasr {

typedef float floatx4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float act_mask(float z) { return (z > 0.f && z < kClamp) ? 1.f : 0.f; }

// ---- dgrad of the grouped node op ------------------------------------------------------------------------------------------
template <int CG, int K, int D>
__global__ __launch_bounds__(256) void grouped_dgrad_kernel(const float* __restrict__ dz, const float* __restrict__ z,
                                                            const float* __restrict__ w, float* __restrict__ dx,
                                                            int channels, int frames, int ld, int groups)
{
    constexpr int LPAD = pad_left(K, D, 1);
    constexpr int SPAN = (K - 1) * D;
    constexpr int L = SPAN - LPAD, R = LPAD;            // the transposed conv reaches L frames back and R ahead
    constexpr int QL = (L + 3) / 4, QR = (R + 3) / 4, NCH = QL + 1 + QR;
    const int nq = ld >> 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = blockIdx.x * 64 + lane;
    const int g = __builtin_amdgcn_readfirstlane(blockIdx.y * 4 + wave);
    const int b = blockIdx.z;
    if (g >= groups) return;
    const bool active = q < nq;
    const size_t row0 = (static_cast<size_t>(b) * channels + static_cast<size_t>(g) * CG) * ld;
    const float* __restrict__ wg = w + static_cast<size_t>(g) * (CG * CG * K);

    float acc[CG][4];
#pragma unroll
    for (int ci = 0; ci < CG; ++ci)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[ci][r] = 0.f;

#pragma unroll 1
    for (int co = 0; co < CG; ++co) {
        const float* __restrict__ drow = dz + row0 + static_cast<size_t>(co) * ld;
        const float* __restrict__ zrow = z + row0 + static_cast<size_t>(co) * ld;
        float gm[NCH * 4];
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int qq = q - QL + c;
            float dv[4] = {0.f, 0.f, 0.f, 0.f}, zv[4] = {0.f, 0.f, 0.f, 0.f};
            if (active && qq >= 0 && qq < nq) { load_frames<4>(drow + qq * 4, dv); load_frames<4>(zrow + qq * 4, zv); }
#pragma unroll
            for (int e = 0; e < 4; ++e) gm[4 * c + e] = dv[e] * act_mask(zv[e]);
        }
#pragma unroll
        for (int j = 0; j < K; ++j) {
#pragma unroll
            for (int ci = 0; ci < CG; ++ci) {
                const float wv = wg[(co * CG + ci) * K + j];
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    acc[ci][r] = __builtin_fmaf(wv, gm[4 * QL + r + LPAD - j * D], acc[ci][r]);
            }
        }
    }
    if (!active) return;
    const int t0 = q * 4;
#pragma unroll
    for (int ci = 0; ci < CG; ++ci) {
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (t0 + r < frames) ? acc[ci][r] : 0.f;
        store_frames<4, false>(dx + row0 + static_cast<size_t>(ci) * ld + t0, o);
    }
}

// ---- wgrad + bias gradient of the grouped node op on the fp32 matrix cores --------------------------------------------------
// One wave per (group, utterance).  v_mfma_f32_16x16x4_f32: lane l supplies A[row l & 15][k = l >> 4] and B[k = l >> 4][col l & 15].
// The 4 k of MFMA m of a 64-frame step are frames t0 + 16 * kq + m (kq = 0..3): a lane then needs 16 CONSECUTIVE frames of its
// row per step -- aligned for the gradient rows, at a tap-dependent offset for the im2col columns.
// Column of tile c: col = 16 c + (l & 15) -> (ci, j) = (col / K, col % K); col == CG * K is the all-ones column (bias gradient).
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));

template <int CG, int K, int D>
__global__ __launch_bounds__(256) void grouped_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dz,
                                                            const float* __restrict__ z, float* __restrict__ part,
                                                            int channels, int frames, int ld, int groups, int batch)
{
    constexpr int LPAD = pad_left(K, D, 1);
    constexpr int NCOL = CG * K + 1, NCT = (NCOL + 15) / 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i16 = lane & 15, kq = lane >> 4;
    const int g = blockIdx.x * 4 + wave;
    const int b = blockIdx.y;
    if (g >= groups) return;
    const size_t row0 = (static_cast<size_t>(b) * channels + static_cast<size_t>(g) * CG) * ld;

    floatx4 acc[NCT];
#pragma unroll
    for (int c = 0; c < NCT; ++c) acc[c] = floatx4{0.f, 0.f, 0.f, 0.f};
    // this lane's im2col column per tile: source row and frame offset
    int col_ci[NCT], col_off[NCT];
    bool col_one[NCT], col_ok[NCT];
#pragma unroll
    for (int c = 0; c < NCT; ++c) {
        const int col = 16 * c + i16;
        col_ok[c] = col < CG * K;
        col_one[c] = col == CG * K;
        col_ci[c] = col_ok[c] ? col / K : 0;
        col_off[c] = col_ok[c] ? (col % K) * D - LPAD : 0;
    }
    const float* __restrict__ drow = dz + row0 + static_cast<size_t>(i16 < CG ? i16 : 0) * ld;
    const float* __restrict__ zrow = z + row0 + static_cast<size_t>(i16 < CG ? i16 : 0) * ld;

    for (int t0 = 0; t0 < frames; t0 += 64) {
        const int ta = t0 + 16 * kq;                        // this lane's 16 frames of the step
        float a[16];
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            float dv[4] = {0.f, 0.f, 0.f, 0.f}, zv[4] = {0.f, 0.f, 0.f, 0.f};
            if (i16 < CG && ta + 4 * h < ld) { load_frames<4>(drow + ta + 4 * h, dv); load_frames<4>(zrow + ta + 4 * h, zv); }
#pragma unroll
            for (int e = 0; e < 4; ++e) a[4 * h + e] = (ta + 4 * h + e < frames) ? dv[e] * act_mask(zv[e]) : 0.f;
        }
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
            float bv[16];
            const int s = ta + col_off[c];
            const float* __restrict__ xrow = x + row0 + static_cast<size_t>(col_ci[c]) * ld;
            if (col_ok[c] && s >= 0 && s + 16 <= frames) {           // interior: four (unaligned) 16-byte loads
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    const f4u v = *reinterpret_cast<const f4u*>(xrow + s + 4 * h);
                    bv[4 * h] = v.x; bv[4 * h + 1] = v.y; bv[4 * h + 2] = v.z; bv[4 * h + 3] = v.w;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int f = s + e;
                    bv[e] = col_one[c] ? 1.f : ((col_ok[c] && f >= 0 && f < frames) ? xrow[f] : 0.f);
                }
            }
#pragma unroll
            for (int m = 0; m < 16; ++m) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], bv[m], acc[c], 0, 0, 0);
        }
    }
    // D[row = co][col]: lane holds rows 4 kq + r of column i16 -> part[(g, b)][tile][co][16]
    float* __restrict__ p = part + (static_cast<size_t>(g) * batch + b) * (NCT * 256);
#pragma unroll
    for (int c = 0; c < NCT; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) p[c * 256 + (4 * kq + r) * 16 + i16] = acc[c][r];
}

// dw[g][co][ci][j], db[g * CG + co] <- sum over utterances of the partials, in utterance order
__global__ __launch_bounds__(256) void grouped_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                                   float* __restrict__ db, int groups, int batch, int cg, int k)
{
    const int ncol = cg * k + 1, nct = (ncol + 15) / 16;
    const int total = groups * cg * ncol;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int col = i % ncol, co = (i / ncol) % cg, g = i / (ncol * cg);
        const float* p = part + static_cast<size_t>(g) * batch * (nct * 256) + (col >> 4) * 256 + co * 16 + (col & 15);
        float s = 0.f;
        for (int b = 0; b < batch; ++b) s += p[static_cast<size_t>(b) * (nct * 256)];
        if (col == cg * k) db[g * cg + co] = s;
        else dw[(static_cast<size_t>(g) * cg + co) * (cg * k) + col] = s;      // (co, ci, j) with col = ci * k + j: torch's layout
    }
}

// ---- LayerNorm backward --------------------------------------------------------------------------------------------------
// Workgroup = 64 frames of one utterance (16 lanes x 4 frames) x 16 channel slots, as the forward kernel.  Pass 1: per frame
// s1 = sum_c g, s2 = sum_c g * xhat (g = dy * gamma) -- the slots' partial sums meet in LDS; pass 2: dx.  The per-channel sums
// over this tile's frames (dgamma, dbeta partials) go to `part` [tile][2][channels]; a second kernel adds the tiles in order.
__global__ __launch_bounds__(256) void layernorm_backward_kernel(const float* __restrict__ x, const float* __restrict__ stats,
                                                                 const float* __restrict__ gamma, const float* __restrict__ dy,
                                                                 float* __restrict__ dx, float* __restrict__ part,
                                                                 int channels, int frames, int ld)
{
    __shared__ float s_a[16][64], s_b[16][64];
    const int ql = threadIdx.x & 15, slot = threadIdx.x >> 4;
    const int nq = ld >> 2;
    const int q = blockIdx.x * 16 + ql;
    const int b = blockIdx.y;
    const bool active = q < nq;
    const size_t base = static_cast<size_t>(b) * channels * ld + static_cast<size_t>(q) * 4;
    float mu[4] = {0.f, 0.f, 0.f, 0.f}, rs[4] = {0.f, 0.f, 0.f, 0.f};
    if (active) {
        const float* st = stats + static_cast<size_t>(b) * 2 * ld + q * 4;
        load_frames<4>(st, mu);
        load_frames<4>(st + ld, rs);
    }
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    const int tile = blockIdx.y * gridDim.x + blockIdx.x;
    float* __restrict__ pg = part + static_cast<size_t>(tile) * 2 * channels;
    for (int c = slot; c < channels; c += 16) {
        float xv[4] = {0.f, 0.f, 0.f, 0.f}, dv[4] = {0.f, 0.f, 0.f, 0.f};
        if (active) { load_frames<4>(x + base + static_cast<size_t>(c) * ld, xv); load_frames<4>(dy + base + static_cast<size_t>(c) * ld, dv); }
        const float gm = gamma[c];
        float pgam = 0.f, pbet = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float xh = (xv[r] - mu[r]) * rs[r];          // rstd == 0 beyond the utterance: xhat = 0, and dy is 0 there
            const float gg = dv[r] * gm;
            s1[r] += gg;
            s2[r] = __builtin_fmaf(gg, xh, s2[r]);
            pgam = __builtin_fmaf(dv[r], xh, pgam);
            pbet += dv[r];
        }
        // the 16 lanes of a slot hold the 64 frames of this channel: reduce over them (xor 1, 2, 4, 8 stays inside the 16-lane row)
#pragma unroll
        for (int d = 8; d >= 1; d >>= 1) { pgam += __shfl_xor(pgam, d); pbet += __shfl_xor(pbet, d); }
        if (ql == 0) { pg[c] = pgam; pg[channels + c] = pbet; }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) { s_a[slot][ql * 4 + r] = s1[r]; s_b[slot][ql * 4 + r] = s2[r]; }
    __syncthreads();
    float m1[4], m2[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float u = 0.f, v = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) { u += s_a[k][ql * 4 + r]; v += s_b[k][ql * 4 + r]; }
        m1[r] = u / static_cast<float>(channels);
        m2[r] = v / static_cast<float>(channels);
    }
    if (!active) return;
    const int t0 = q * 4;
    for (int c = slot; c < channels; c += 16) {
        float xv[4], dv[4], o[4];
        load_frames<4>(x + base + static_cast<size_t>(c) * ld, xv);
        load_frames<4>(dy + base + static_cast<size_t>(c) * ld, dv);
        const float gm = gamma[c];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float xh = (xv[r] - mu[r]) * rs[r];
            o[r] = (t0 + r < frames) ? rs[r] * (dv[r] * gm - m1[r] - xh * m2[r]) : 0.f;
        }
        store_frames<4, false>(dx + base + static_cast<size_t>(c) * ld, o);
    }
}

// 64 channels x 4 slices of the tile list per workgroup (a slice takes tiles s, s + 4, ...), the slices meet in LDS in a fixed order:
// bit-reproducible.  (One thread per channel walking all 1 024 tiles of a 64 x 1000 batch ran on 5 workgroups: 250 us per LayerNorm.)
__global__ __launch_bounds__(256) void layernorm_backward_reduce_kernel(const float* __restrict__ part, float* __restrict__ dgamma,
                                                                        float* __restrict__ dbeta, int tiles, int channels)
{
    __shared__ float s_a[4][64], s_b[4][64];
    const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    float a[4] = {0.f, 0.f, 0.f, 0.f}, bsum[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < channels) {
        int t = slice;
        for (; t + 12 < tiles; t += 16) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const size_t at = static_cast<size_t>(t + 4 * u) * 2 * channels + c;
                a[u] += part[at];
                bsum[u] += part[at + channels];
            }
        }
        for (; t < tiles; t += 4) { const size_t at = static_cast<size_t>(t) * 2 * channels + c; a[0] += part[at]; bsum[0] += part[at + channels]; }
    }
    s_a[slice][lane] = (a[0] + a[1]) + (a[2] + a[3]);
    s_b[slice][lane] = (bsum[0] + bsum[1]) + (bsum[2] + bsum[3]);
    __syncthreads();
    if (slice == 0 && c < channels) {
        dgamma[c] = (s_a[0][lane] + s_a[1][lane]) + (s_a[2][lane] + s_a[3][lane]);
        dbeta[c] = (s_b[0][lane] + s_b[1][lane]) + (s_b[2][lane] + s_b[3][lane]);
    }
}

template <int CG>
static int launch_grouped_backward(int kernel, int dilation, const float* x, const float* w, const float* z, const float* dz, float* dx,
                                   float* part, int batch, int channels, int frames, int ld, int groups, hipStream_t s)
{
    const dim3 dgrid((ld / 4 + 63) / 64, (groups + 3) / 4, batch), wgrid((groups + 3) / 4, batch);
#define NBASR_BWD(KK, DD)                                                                                                          \
    do {                                                                                                                           \
        if (dx) hipLaunchKernelGGL((grouped_dgrad_kernel<CG, KK, DD>), dgrid, dim3(256), 0, s, dz, z, w, dx, channels, frames, ld, groups); \
        if (part) hipLaunchKernelGGL((grouped_wgrad_kernel<CG, KK, DD>), wgrid, dim3(256), 0, s, x, dz, z, part, channels, frames, ld, groups, batch); \
        return launch_status("nbasr_grouped_conv1d_backward");                                                                     \
    } while (0)
    if (kernel == 5 && dilation == 1) NBASR_BWD(5, 1);
    if (kernel == 5 && dilation == 2) NBASR_BWD(5, 2);
    if (kernel == 7 && dilation == 1) NBASR_BWD(7, 1);
    if (kernel == 7 && dilation == 2) NBASR_BWD(7, 2);
#undef NBASR_BWD
    set_error("nbasr_grouped_conv1d_backward: unsupported (kernel=%d, dilation=%d)", kernel, dilation);
    return NBASR_EINVAL;
}

}  // namespace nbasr

using namespace nbasr;

extern "C" size_t nbasr_grouped_conv1d_backward_workspace_bytes(int batch, int channels, int groups, int kernel)
{
    if (batch <= 0 || channels <= 0 || groups <= 0 || channels % groups || kernel <= 0) return 0;
    const int ncol = (channels / groups) * kernel + 1;
    return static_cast<size_t>(groups) * batch * ((ncol + 15) / 16) * 256 * sizeof(float);
}

extern "C" int nbasr_grouped_conv1d_backward(const float* x, const float* w, const float* z, const float* dz, float* dx, float* dw,
                                             float* db, float* workspace, int batch, int channels, int frames, int ld, int groups,
                                             int kernel, int dilation, nbasr_stream_t stream)
{
    clear_error();
    NBASR_REQUIRE(batch >= 0 && channels > 0 && frames >= 0 && groups > 0 && channels % groups == 0, NBASR_EINVAL,
                  "nbasr_grouped_conv1d_backward: bad sizes");
    NBASR_REQUIRE(ld >= frames && ld % 4 == 0, NBASR_EALIGN, "nbasr_grouped_conv1d_backward: ld=%d must be >= frames=%d and a multiple of 4", ld, frames);
    NBASR_REQUIRE((dw == nullptr) == (db == nullptr), NBASR_ENULL, "nbasr_grouped_conv1d_backward: dw and db come together");
    NBASR_REQUIRE(!dw || workspace, NBASR_ENULL, "nbasr_grouped_conv1d_backward: the weight gradient needs the workspace");
    if (batch == 0 || ld == 0) {
        if (dw) {
            (void)hipMemsetAsync(dw, 0, sizeof(float) * channels * (channels / groups) * kernel, as_stream(stream));
            (void)hipMemsetAsync(db, 0, sizeof(float) * channels, as_stream(stream));
        }
        return NBASR_OK;
    }
    NBASR_REQUIRE(w && z && dz && (dx || dw), NBASR_ENULL, "nbasr_grouped_conv1d_backward: w, z, dz and at least one output must be non-NULL");
    NBASR_REQUIRE(!dw || x, NBASR_ENULL, "nbasr_grouped_conv1d_backward: the weight gradient needs x");
    NBASR_REQUIRE(aligned16(x) && aligned16(z) && aligned16(dz) && aligned16(dx) && aligned16(workspace), NBASR_EALIGN,
                  "nbasr_grouped_conv1d_backward: pointers must be 16-byte aligned");
    NBASR_REQUIRE(batch <= 65535, NBASR_EINVAL, "nbasr_grouped_conv1d_backward: batch %d > 65535", batch);
    hipStream_t s = as_stream(stream);
    float* part = dw ? workspace : nullptr;
    int rc;
    switch (channels / groups) {
        case 6:  rc = launch_grouped_backward<6>(kernel, dilation, x, w, z, dz, dx, part, batch, channels, frames, ld, groups, s); break;
        case 8:  rc = launch_grouped_backward<8>(kernel, dilation, x, w, z, dz, dx, part, batch, channels, frames, ld, groups, s); break;
        case 10: rc = launch_grouped_backward<10>(kernel, dilation, x, w, z, dz, dx, part, batch, channels, frames, ld, groups, s); break;
        case 12: rc = launch_grouped_backward<12>(kernel, dilation, x, w, z, dz, dx, part, batch, channels, frames, ld, groups, s); break;
        default:
            set_error("nbasr_grouped_conv1d_backward: channels/groups=%d unsupported (model widths give 6, 8, 10, 12)", channels / groups);
            return NBASR_EINVAL;
    }
    if (rc != NBASR_OK || !dw) return rc;
    const int total = channels * ((channels / groups) * kernel + 1);
    hipLaunchKernelGGL(grouped_wgrad_reduce_kernel, dim3((total + 255) / 256), dim3(256), 0, s, part, dw, db, groups, batch, channels / groups, kernel);
    return launch_status("nbasr_grouped_conv1d_backward");
}

extern "C" size_t nbasr_layernorm_backward_workspace_bytes(int batch, int channels, int ld)
{
    if (batch <= 0 || channels <= 0 || ld <= 0) return 0;
    return static_cast<size_t>(batch) * ((ld / 4 + 15) / 16) * 2 * channels * sizeof(float);
}

extern "C" int nbasr_layernorm_channels_backward(const float* x, const float* stats, const float* gamma, const float* dy, float* dx,
                                                 float* dgamma, float* dbeta, float* workspace, int batch, int channels, int frames,
                                                 int ld, nbasr_stream_t stream)
{
    clear_error();
    NBASR_REQUIRE(batch >= 0 && channels > 0 && frames >= 0, NBASR_EINVAL, "nbasr_layernorm_channels_backward: bad sizes");
    NBASR_REQUIRE(ld >= frames && ld % 4 == 0, NBASR_EALIGN, "nbasr_layernorm_channels_backward: ld=%d must be >= frames=%d and a multiple of 4", ld, frames);
    NBASR_REQUIRE(dgamma && dbeta, NBASR_ENULL, "nbasr_layernorm_channels_backward: dgamma / dbeta are NULL");
    if (batch == 0 || ld == 0) {
        (void)hipMemsetAsync(dgamma, 0, sizeof(float) * channels, as_stream(stream));
        (void)hipMemsetAsync(dbeta, 0, sizeof(float) * channels, as_stream(stream));
        return NBASR_OK;
    }
    NBASR_REQUIRE(x && stats && gamma && dy && dx && workspace, NBASR_ENULL, "nbasr_layernorm_channels_backward: NULL pointer");
    NBASR_REQUIRE(aligned16(x) && aligned16(stats) && aligned16(dy) && aligned16(dx), NBASR_EALIGN,
                  "nbasr_layernorm_channels_backward: pointers must be 16-byte aligned");
    NBASR_REQUIRE(batch <= 65535, NBASR_EINVAL, "nbasr_layernorm_channels_backward: batch %d > 65535", batch);
    const int tiles_x = (ld / 4 + 15) / 16;
    hipLaunchKernelGGL(layernorm_backward_kernel, dim3(tiles_x, batch), dim3(256), 0, as_stream(stream), x, stats, gamma, dy, dx, workspace,
                       channels, frames, ld);
    hipLaunchKernelGGL(layernorm_backward_reduce_kernel, dim3((channels + 63) / 64), dim3(256), 0, as_stream(stream), workspace, dgamma, dbeta,
                       tiles_x * batch, channels);
    return launch_status("nbasr_layernorm_channels_backward");
}

// ---- building blocks of the GEMM-shaped backward passes (linear node op, head, dense downsample convs) -------------------------
// The products themselves run on the exact-fp32 MFMA GEMMs that exist for the forward (nbasr_pointwise_linear,
// nbasr_dense_conv1d_linear); these kernels put the operands into the layouts those GEMMs read.
namespace nbasr {

// dz = dy where 0 < y < 20, else 0  (y = min(relu(z), 20): the gradient passes where neither the ReLU nor the clamp is active)
__global__ __launch_bounds__(256) void relu_clamp_backward_kernel(const float4* __restrict__ y, const float4* __restrict__ dy,
                                                                   float4* __restrict__ dz, size_t n4)
{
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 a = y[i], g = dy[i];
        dz[i] = make_float4((a.x > 0.f && a.x < kClamp) ? g.x : 0.f, (a.y > 0.f && a.y < kClamp) ? g.y : 0.f,
                            (a.z > 0.f && a.z < kClamp) ? g.z : 0.f, (a.w > 0.f && a.w < kClamp) ? g.w : 0.f);
    }
}

// The weight gradient of a k-tap convolution is a GEMM over (utterance, output frame):
//     dw[co][ci][j] = sum_{b,t} dz[b][co][t] * xpad[b][ci][t * stride + j - lpad]
// cols[(b * t_pad + t)][ci * taps + j] = xpad[b][ci][t * stride + j - lpad]   (0 outside the utterance and for t >= frames_out),
// with one extra column of ones per row (index c_in * taps, then zeros up to ld_cols) so that the same GEMM yields the bias
// gradient.  One thread per (row, column): reads are strided, the writes of a wave are contiguous.
__global__ __launch_bounds__(256) void conv_cols_kernel(const float* __restrict__ x, float* __restrict__ cols, int c_in, int frames_in,
                                                        int ld_in, int frames_out, int t_pad, int taps, int stride, int lpad, int ld_cols)
{
    const int n = blockIdx.x * blockDim.x + threadIdx.x;           // column
    const int row = blockIdx.y, b = blockIdx.z;                    // output frame (padded), utterance
    if (n >= ld_cols) return;
    float v = 0.f;
    const int ncols = c_in * taps;
    if (row < frames_out) {
        if (n < ncols) {
            const int ci = n / taps, j = n - ci * taps;
            const int u = row * stride + j - lpad;
            if (u >= 0 && u < frames_in) v = x[(static_cast<size_t>(b) * c_in + ci) * ld_in + u];
        } else if (n == ncols) {
            v = 1.f;
        }
    }
    cols[(static_cast<size_t>(b) * t_pad + row) * ld_cols + n] = v;
}

// rows[co][b * t_pad + t] = dz[b][co][t] (0 for t >= frames): the other operand of that GEMM, "weights" of (c_out, batch * t_pad)
__global__ __launch_bounds__(256) void rows_of_channels_kernel(const float* __restrict__ dz, float* __restrict__ rows, int batch, int channels,
                                                               int frames, int ld, int t_pad)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int co = blockIdx.y, b = blockIdx.z;
    if (t >= t_pad) return;
    rows[static_cast<size_t>(co) * batch * t_pad + static_cast<size_t>(b) * t_pad + t] =
        t < frames ? dz[(static_cast<size_t>(b) * channels + co) * ld + t] : 0.f;
}

// The input gradient of a strided convolution is a stride-1 convolution of the zero-stuffed output gradient with the flipped,
// channel-transposed kernel: up[b][co][shift + t * stride] = dz[b][co][t], zeros elsewhere (frames_up columns, pitch ld_up).
__global__ __launch_bounds__(256) void zero_stuff_kernel(const float* __restrict__ dz, float* __restrict__ up, int rows, int frames, int ld,
                                                         int frames_up, int ld_up, int stride, int shift)
{
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= ld_up) return;
    const int s = u - shift;
    const bool hit = u < frames_up && s >= 0 && s % stride == 0 && s / stride < frames;
    for (int r = blockIdx.y; r < rows; r += gridDim.y)
        up[static_cast<size_t>(r) * ld_up + u] = hit ? dz[static_cast<size_t>(r) * ld + s / stride] : 0.f;
}

// The same input gradient as ONE GEMM over the un-stuffed gradient plus a fold: cols (T', B, C_in * 8), time-major -- the store of the
// split 16-bit GEMM nbasr_lstm_input_projection_packed run on w^T (rows (ci, tap), K = C_out) and the masked output gradient -- holds
// every tap's contribution; dx[b][ci][u] = sum over taps j with (u + lpad - j) = t * stride, 0 <= t < T', of cols[t][b][ci * 8 + j],
// taps in ascending order (fixed summation order).  Half the products of the zero-stuffed conv at stride 2.  One workgroup: one
// utterance x 32 channels x 32 frames; the <= 39 cols rows it needs (1 KiB each, contiguous) go through LDS so that HBM is read in
// whole rows and written in 128-byte runs.
constexpr int FOLD_T = 32, FOLD_C = 32, FOLD_TAPS = 8, FOLD_LD = FOLD_C * FOLD_TAPS + 4;
template <int S>
__global__ __launch_bounds__(256) void conv_fold_kernel(const float* __restrict__ cols, float* __restrict__ dx, int batch, int c_in,
                                                        int frames_in, int ld_in, int frames_out, int lpad)
{
    constexpr int ROWS = (FOLD_T + FOLD_TAPS - 2) / S + 2;
    __shared__ __attribute__((aligned(16))) float s_cols[ROWS][FOLD_LD];
    const int u0 = blockIdx.x * FOLD_T, ci0 = blockIdx.y * FOLD_C, b = blockIdx.z;
    const int n_lo = u0 + lpad - (FOLD_TAPS - 1);
    const int t_lo = n_lo > 0 ? (n_lo + S - 1) / S : 0;
    const int t_hi = min(frames_out - 1, (u0 + FOLD_T - 1 + lpad) / S);
    const int row_floats = c_in * FOLD_TAPS;
    {
        const int q = threadIdx.x & 63, r0 = threadIdx.x >> 6;
        const int col = ci0 * FOLD_TAPS + q * 4;
        for (int r = r0; t_lo + r <= t_hi; r += 4) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (col < row_floats)
                v = *reinterpret_cast<const float4*>(cols + (static_cast<size_t>(t_lo + r) * batch + b) * row_floats + col);
            *reinterpret_cast<float4*>(&s_cols[r][q * 4]) = v;
        }
    }
    __syncthreads();
    const int u = u0 + (threadIdx.x & (FOLD_T - 1));
    if (u >= ld_in) return;
    for (int cl = threadIdx.x / FOLD_T; cl < FOLD_C; cl += 256 / FOLD_T) {
        const int ci = ci0 + cl;
        if (ci >= c_in) break;
        float sum = 0.f;
        if (u < frames_in) {
#pragma unroll
            for (int j = 0; j < FOLD_TAPS; ++j) {
                const int n = u + lpad - j;
                if (n >= 0 && n % S == 0 && n / S <= t_hi) sum += s_cols[n / S - t_lo][cl * FOLD_TAPS + j];
            }
        }
        dx[(static_cast<size_t>(b) * c_in + ci) * ld_in + u] = sum;           // pitch columns: exact zeros
    }
}

}  // namespace nbasr

extern "C" int nbasr_relu_clamp_backward(const float* y, const float* dy, float* dz, long long n, nbasr_stream_t stream)
{
    clear_error();
    NBASR_REQUIRE(n >= 0 && n % 4 == 0, NBASR_EINVAL, "nbasr_relu_clamp_backward: n=%lld must be a non-negative multiple of 4", n);
    if (n == 0) return NBASR_OK;
    NBASR_REQUIRE(y && dy && dz, NBASR_ENULL, "nbasr_relu_clamp_backward: NULL pointer");
    NBASR_REQUIRE(aligned16(y) && aligned16(dy) && aligned16(dz), NBASR_EALIGN, "nbasr_relu_clamp_backward: pointers must be 16-byte aligned");
    const size_t n4 = static_cast<size_t>(n / 4);
    const unsigned grid = static_cast<unsigned>(n4 / 256 + 1 < 4096 ? n4 / 256 + 1 : 4096);
    hipLaunchKernelGGL(relu_clamp_backward_kernel, dim3(grid), dim3(256), 0, as_stream(stream), reinterpret_cast<const float4*>(y),
                       reinterpret_cast<const float4*>(dy), reinterpret_cast<float4*>(dz), n4);
    return launch_status("nbasr_relu_clamp_backward");
}

extern "C" int nbasr_conv_cols(const float* x, float* cols, int batch, int c_in, int frames_in, int ld_in, int frames_out, int t_pad,
                               int taps, int stride, int lpad, int ld_cols, nbasr_stream_t stream)
{
    clear_error();
    NBASR_REQUIRE(batch >= 0 && c_in > 0 && frames_in >= 0 && ld_in >= frames_in && frames_out >= 0 && t_pad >= frames_out && taps > 0 &&
                  stride > 0 && lpad >= 0 && ld_cols >= c_in * taps + 1, NBASR_EINVAL, "nbasr_conv_cols: bad sizes");
    if (batch == 0 || t_pad == 0) return NBASR_OK;
    NBASR_REQUIRE(x && cols, NBASR_ENULL, "nbasr_conv_cols: NULL pointer");
    NBASR_REQUIRE(batch <= 65535 && t_pad <= 65535, NBASR_EINVAL, "nbasr_conv_cols: batch / frames > 65535");
    hipLaunchKernelGGL(conv_cols_kernel, dim3((ld_cols + 255) / 256, t_pad, batch), dim3(256), 0, as_stream(stream), x, cols, c_in, frames_in,
                       ld_in, frames_out, t_pad, taps, stride, lpad, ld_cols);
    return launch_status("nbasr_conv_cols");
}

extern "C" int nbasr_rows_of_channels(const float* dz, float* rows, int batch, int channels, int frames, int ld, int t_pad,
                                      nbasr_stream_t stream)
{
    clear_error();
    NBASR_REQUIRE(batch >= 0 && channels > 0 && frames >= 0 && ld >= frames && t_pad >= frames, NBASR_EINVAL, "nbasr_rows_of_channels: bad sizes");
    if (batch == 0 || t_pad == 0) return NBASR_OK;
    NBASR_REQUIRE(dz && rows, NBASR_ENULL, "nbasr_rows_of_channels: NULL pointer");
    NBASR_REQUIRE(batch <= 65535 && channels <= 65535, NBASR_EINVAL, "nbasr_rows_of_channels: batch / channels > 65535");
    hipLaunchKernelGGL(rows_of_channels_kernel, dim3((t_pad + 255) / 256, channels, batch), dim3(256), 0, as_stream(stream), dz, rows, batch,
                       channels, frames, ld, t_pad);
    return launch_status("nbasr_rows_of_channels");
}

extern "C" int nbasr_zero_stuff(const float* dz, float* up, int rows, int frames, int ld, int frames_up, int ld_up, int stride, int shift,
                                nbasr_stream_t stream)
{
    clear_error();
    NBASR_REQUIRE(rows >= 0 && frames >= 0 && ld >= frames && frames_up >= 0 && ld_up >= frames_up && stride > 0 && shift >= 0, NBASR_EINVAL,
                  "nbasr_zero_stuff: bad sizes");
    if (rows == 0 || ld_up == 0) return NBASR_OK;
    NBASR_REQUIRE(dz && up, NBASR_ENULL, "nbasr_zero_stuff: NULL pointer");
    hipLaunchKernelGGL(zero_stuff_kernel, dim3((ld_up + 255) / 256, rows < 4096 ? rows : 4096), dim3(256), 0, as_stream(stream), dz, up, rows,
                       frames, ld, frames_up, ld_up, stride, shift);
    return launch_status("nbasr_zero_stuff");
}

extern "C" int nbasr_conv_fold(const float* cols, float* dx, int batch, int c_in, int frames_in, int ld_in, int frames_out, int taps,
                               int stride, int lpad, nbasr_stream_t stream)
{
    clear_error();
    NBASR_REQUIRE(batch >= 0 && c_in > 0 && frames_in >= 0 && ld_in >= frames_in && frames_out >= 0 && lpad >= 0, NBASR_EINVAL,
                  "nbasr_conv_fold: bad sizes");
    NBASR_REQUIRE(taps == FOLD_TAPS && (stride == 1 || stride == 2), NBASR_EINVAL, "nbasr_conv_fold: taps=%d stride=%d (8 taps, stride 1 | 2)",
                  taps, stride);
    NBASR_REQUIRE(batch <= 65535 && (c_in + FOLD_C - 1) / FOLD_C <= 65535, NBASR_EINVAL, "nbasr_conv_fold: batch=%d / c_in=%d beyond the grid", batch, c_in);
    if (batch == 0 || ld_in == 0) return NBASR_OK;
    NBASR_REQUIRE(cols && dx, NBASR_ENULL, "nbasr_conv_fold: NULL pointer");
    NBASR_REQUIRE(aligned16(cols), NBASR_EALIGN, "nbasr_conv_fold: cols must be 16-byte aligned");
    const dim3 grid((ld_in + FOLD_T - 1) / FOLD_T, (c_in + FOLD_C - 1) / FOLD_C, batch);
    if (stride == 1)
        hipLaunchKernelGGL(conv_fold_kernel<1>, grid, dim3(256), 0, as_stream(stream), cols, dx, batch, c_in, frames_in, ld_in, frames_out, lpad);
    else
        hipLaunchKernelGGL(conv_fold_kernel<2>, grid, dim3(256), 0, as_stream(stream), cols, dx, batch, c_in, frames_in, ld_in, frames_out, lpad);
    return launch_status("nbasr_conv_fold");
}

// ---- LSTM backward (BPTT), correctness first -----------------------------------------------------------------------------------------
// Layout of everything below: (rows, frames, ldb) with the utterances innermost (ldb = batch rounded up to 4), rows = 4H for the
// gate tensors (PyTorch order i, f, g, o) and H for the cell states -- the layout in which the per-step GEMM
// dh_(t-1) += w_hh^T . dpre_t and the three batched GEMMs of the weight / input gradients read their operands directly.
namespace nbasr {

__device__ __forceinline__ float sigmoid_bw(float v) { return 1.0f / (1.0f + expf(-v)); }

// pre (4H, T, ldb): gate pre-activations of ALL frames (input projection + w_hh . h_(t-1), recomputed from the saved h by one
// GEMM) -> overwritten by the gate activations; cells (H, T, ldb) <- c_t.  One thread per (unit, utterance), a serial scan over t.
__global__ __launch_bounds__(256) void lstm_gate_scan_kernel(float* __restrict__ pre, float* __restrict__ cells, int hidden, int frames,
                                                             int batch, int ldb)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y;
    if (b >= batch) return;
    const size_t plane = static_cast<size_t>(frames) * ldb, gate_stride = static_cast<size_t>(hidden) * plane;
    float c = 0.f;
    for (int t = 0; t < frames; ++t) {
        const size_t at = static_cast<size_t>(j) * plane + static_cast<size_t>(t) * ldb + b;
        const float i = sigmoid_bw(pre[at]), f = sigmoid_bw(pre[at + gate_stride]), g = tanhf(pre[at + 2 * gate_stride]),
                    o = sigmoid_bw(pre[at + 3 * gate_stride]);
        c = f * c + i * g;
        pre[at] = i; pre[at + gate_stride] = f; pre[at + 2 * gate_stride] = g; pre[at + 3 * gate_stride] = o;
        cells[at] = c;
    }
}

// One step of the reverse recurrence: dh_out (H, T, ldb) = dL/dh of the layer's output, w_hh_t (H, 4H) the transposed recurrent weight,
// dc (H, ldb) the carried dL/dc_t part -> dpre[:, t, :] (4H rows; reads dpre[:, t + 1, :]) and the updated carry dL/dc_(t-1).
constexpr int LSTM_BW_UNITS = 4;                              // hidden units per workgroup: they share every load of dpre[:, t + 1, :]
constexpr int LSTM_BW_SLICES = 16;                            // waves per workgroup: each sums a sixteenth of the 4H gate rows
constexpr int LSTM_BW_DEPTH = 16;                             // loads in flight per wave
__global__ __launch_bounds__(64 * LSTM_BW_SLICES) void lstm_backward_step_kernel(const float* __restrict__ dh_out, const float* __restrict__ w_hh_t,
                                                                 float* __restrict__ dc, const float* __restrict__ acts,
                                                                 const float* __restrict__ cells, float* __restrict__ dpre, int hidden,
                                                                 int frames, int batch, int ldb, int t)
{
    // 1024 threads = 64 utterances x 16 slices of the 4H gate rows: each slice (one wave) sums its part of
    // (w_hh^T . dpre[:, t + 1, :])[j][b] for the workgroup's 4 units j -- the recurrent weights are wave-uniform scalars, 16 loads of
    // dpre in flight -- the slices meet in LDS, slice 0 does the cell arithmetic.  The step is bound by load LATENCY, not bandwidth
    // (512 KB of dpre[:, t + 1, :] out of L2): as a launch of the tiled GEMM this (500 x 2000) x (2000 x 64) product ran on 4
    // workgroups, 240 us per frame; one unit per workgroup, 4 slices, 8 loads in flight -- 62 dependent round trips -- 30 us.
    __shared__ float s_part[LSTM_BW_SLICES][LSTM_BW_UNITS][64];
    const int lane = threadIdx.x & 63, slice = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);    // wave-uniform: scalar weight loads
    const int b = blockIdx.x * 64 + lane;
    const int j0 = blockIdx.y * LSTM_BW_UNITS;
    const size_t plane = static_cast<size_t>(frames) * ldb, gate_stride = static_cast<size_t>(hidden) * plane;
    float rec[LSTM_BW_UNITS];
#pragma unroll
    for (int q = 0; q < LSTM_BW_UNITS; ++q) rec[q] = 0.f;
    if (t + 1 < frames && b < ldb) {
        // 4H is a multiple of 16 (the entry point checks hidden % 4): slices of a multiple of 16 rows leave no tail
        const int rows = 4 * hidden, per = ((rows + LSTM_BW_SLICES - 1) / LSTM_BW_SLICES + LSTM_BW_DEPTH - 1) / LSTM_BW_DEPTH * LSTM_BW_DEPTH;
        const int g0 = min(rows, slice * per), g1 = min(rows, g0 + per);
        const float* __restrict__ wrow = w_hh_t + static_cast<size_t>(j0) * rows;
        const float* __restrict__ dnext = dpre + static_cast<size_t>(t + 1) * ldb + b;
        float r[LSTM_BW_UNITS][2];
#pragma unroll
        for (int q = 0; q < LSTM_BW_UNITS; ++q) r[q][0] = r[q][1] = 0.f;
        for (int g = g0; g < g1; g += LSTM_BW_DEPTH) {
            float d[LSTM_BW_DEPTH];
#pragma unroll
            for (int u = 0; u < LSTM_BW_DEPTH; ++u) d[u] = dnext[static_cast<size_t>(g + u) * plane];
#pragma unroll
            for (int q = 0; q < LSTM_BW_UNITS; ++q)
#pragma unroll
                for (int u = 0; u < LSTM_BW_DEPTH; ++u) r[q][u & 1] = __builtin_fmaf(wrow[static_cast<size_t>(q) * rows + g + u], d[u], r[q][u & 1]);
        }
#pragma unroll
        for (int q = 0; q < LSTM_BW_UNITS; ++q) rec[q] = r[q][0] + r[q][1];
    }
#pragma unroll
    for (int q = 0; q < LSTM_BW_UNITS; ++q) s_part[slice][q][lane] = rec[q];
    __syncthreads();
    if (slice != 0 || b >= ldb) return;
#pragma unroll
    for (int q = 0; q < LSTM_BW_UNITS; ++q) {
        const int j = j0 + q;
        const size_t at = static_cast<size_t>(j) * plane + static_cast<size_t>(t) * ldb + b;
        const size_t hb = static_cast<size_t>(j) * ldb + b;
        if (b >= batch) {                                      // pitch columns: exact zeros for the GEMMs that read them
            dpre[at] = 0.f; dpre[at + gate_stride] = 0.f; dpre[at + 2 * gate_stride] = 0.f; dpre[at + 3 * gate_stride] = 0.f;
            continue;
        }
        const float i = acts[at], f = acts[at + gate_stride], g = acts[at + 2 * gate_stride], o = acts[at + 3 * gate_stride];
        const float c = cells[at], c_prev = t > 0 ? cells[at - ldb] : 0.f;
        const float tc = tanhf(c);
        float through = 0.f;                                   // from frame t + 1 through the recurrence, slices in ascending order
#pragma unroll
        for (int sl = 0; sl < LSTM_BW_SLICES; ++sl) through += s_part[sl][q][lane];
        const float gh = dh_out[at] + through;                 // dL/dh_t: from the layer's output + the recurrence
        const float d_o = gh * tc;
        const float d_c = dc[hb] + gh * o * (1.f - tc * tc);
        dpre[at] = d_c * g * i * (1.f - i);
        dpre[at + gate_stride] = d_c * c_prev * f * (1.f - f);
        dpre[at + 2 * gate_stride] = d_c * i * (1.f - g * g);
        dpre[at + 3 * gate_stride] = d_o * o * (1.f - o);
        dc[hb] = d_c * f;
    }
}

}  // namespace nbasr

extern "C" int nbasr_lstm_gate_scan(float* pre, float* cells, int hidden, int frames, int batch, int ldb, nbasr_stream_t stream)
{
    clear_error();
    NBASR_REQUIRE(hidden > 0 && frames >= 0 && batch >= 0 && ldb >= batch, NBASR_EINVAL, "nbasr_lstm_gate_scan: bad sizes");
    if (frames == 0 || batch == 0) return NBASR_OK;
    NBASR_REQUIRE(pre && cells, NBASR_ENULL, "nbasr_lstm_gate_scan: NULL pointer");
    NBASR_REQUIRE(hidden <= 65535, NBASR_EINVAL, "nbasr_lstm_gate_scan: hidden %d > 65535", hidden);
    hipLaunchKernelGGL(lstm_gate_scan_kernel, dim3((batch + 63) / 64, hidden), dim3(64), 0, as_stream(stream), pre, cells, hidden, frames, batch, ldb);
    return launch_status("nbasr_lstm_gate_scan");
}

extern "C" int nbasr_lstm_backward_step(const float* dh_out, const float* w_hh_t, float* dc, const float* acts, const float* cells, float* dpre,
                                        int hidden, int frames, int batch, int ldb, int t, nbasr_stream_t stream)
{
    clear_error();
    NBASR_REQUIRE(hidden > 0 && frames > 0 && batch > 0 && ldb >= batch && t >= -1 && t < frames, NBASR_EINVAL, "nbasr_lstm_backward_step: bad sizes");
    NBASR_REQUIRE(dh_out && w_hh_t && dc && acts && cells && dpre, NBASR_ENULL, "nbasr_lstm_backward_step: NULL pointer");
    NBASR_REQUIRE(hidden % LSTM_BW_UNITS == 0 && hidden / LSTM_BW_UNITS <= 65535, NBASR_EINVAL, "nbasr_lstm_backward_step: hidden %d (a multiple of 4, at most 262140)", hidden);
    if (t == -1) {
        // the whole reverse recurrence, frames T-1 .. 0, as one chain of launches (round 6: the trainer's step issued them from a python
        // loop); replayed as one cached graph where the same buffers recur (common.h)
        struct Ctx { hipStream_t s; const float* dh; const float* w; float* dc; const float* acts; const float* cells; float* dpre; int hidden, frames, batch, ldb; };
        Ctx ctx{as_stream(stream), dh_out, w_hh_t, dc, acts, cells, dpre, hidden, frames, batch, ldb};
        const ChainKey key{{dh_out, w_hh_t, dc, acts, dpre}, {hidden, frames, batch, ldb, -32}};
        return replay_chain(ctx.s, key, "nbasr_lstm_backward_step", [](void* p) {
            const Ctx& c = *static_cast<const Ctx*>(p);
            for (int u = c.frames - 1; u >= 0; --u)
                hipLaunchKernelGGL(lstm_backward_step_kernel, dim3((c.ldb + 63) / 64, c.hidden / LSTM_BW_UNITS), dim3(64 * LSTM_BW_SLICES), 0, c.s, c.dh, c.w, c.dc, c.acts,
                                   c.cells, c.dpre, c.hidden, c.frames, c.batch, c.ldb, u);
        }, &ctx);
    }
    hipLaunchKernelGGL(lstm_backward_step_kernel, dim3((ldb + 63) / 64, hidden / LSTM_BW_UNITS), dim3(64 * LSTM_BW_SLICES), 0, as_stream(stream), dh_out, w_hh_t, dc, acts, cells, dpre,
                       hidden, frames, batch, ldb, t);
    return launch_status("nbasr_lstm_backward_step");
}
