// Per-frame linear maps of the bf16 storage path (BASELINE config 4: model.to(torch.bfloat16)) on the bf16 matrix cores, ONE MFMA per
// product (round 4).  Rounds 2-3 bridged both users through the fp32 kernels of gemm_pointwise_split.hip -- the
// activation converted (or normalised) to fp32, split into two fp16 terms, three MFMAs per product:
//   * the LSTM input projection (reference model.py:100,118-121 under model.to(bfloat16)): x = the encoder output, a bfloat16 tensor
//     with its LayerNorm still pending; gates (frames, batch, 4 H) fp32, time-major;
//   * the `linear` node op of a search cell (reference ops.py:42-50: permute, Linear, ReLU, min(20), permute) + the node's skip sum
//     (model.py:13-22) on bf16 rows.
// Every tensor of the bf16 model IS a bfloat16 tensor and so is every weight, so the products go to v_mfma_f32_16x16x32_bf16 unchanged:
// exact bf16 x bf16 products, fp32 accumulation -- the reference's arithmetic up to the order of the sums; the normalised input is
// rounded to bf16 exactly where the reference's LayerNorm module rounds its output.
//
//   image:  x (batch, C, ld) bf16 [+ pending LayerNorm]  ->  [b][frame tile of 256][K-step of 32 channels]
//           [16-channel block][8-channel half][256 frames][8 channels] bf16   (16 KiB per (tile, K-step), contiguous); no scales --
//           bf16 has fp32's exponent range;
//   GEMM:   128 x 256 tile per 512-thread workgroup (8 waves of 64 x 64 = 4 x 4 MFMA tiles), K-step = 32 channels; per step 8 KiB of
//           packed weights + 16 KiB of image by LDS-DMA, double-buffered, one barrier per step; frame-tile-major tile order (the row
//           tiles of a frame tile share its image through an XCD's L2); epilogue: bias [+ second bias], [ReLU, min(20)], [skip sum with
//           the pending LayerNorm on skip0], ONE rounding to bf16 -- or the fp32 time-major store of the LSTM gates.
#include "storage.h"

namespace nbasr {

typedef float pwb_f4 __attribute__((ext_vector_type(4)));
typedef __bf16 pwb_bf8 __attribute__((ext_vector_type(8)));
typedef unsigned pwb_u4 __attribute__((ext_vector_type(4)));

constexpr int PWB_M = 128, PWB_N = 256, PWB_K = 32;
constexpr int PWB_THREADS = 512;
constexpr int PWB_A_STEP = 2 * 2 * PWB_M * 16;                  // [block][half][128 rows][8 ch] bf16 = 8 KiB
constexpr int PWB_X_STEP = 2 * 2 * PWB_N * 16;                  // [block][half][256 frames][8 ch] bf16 = 16 KiB
constexpr int PWB_LDS = 2 * (PWB_A_STEP + PWB_X_STEP);          // 48 KiB

// weights (c_out, c_in) fp32 (the values of a bf16 parameter: the rounding is exact) -> [row tile][K-step][block][half][128 rows][8 ch]
__global__ __launch_bounds__(256) void pwb_pack_weights_kernel(const float* __restrict__ w, unsigned short* __restrict__ wp, int c_out, int c_in,
                                                               int n_mt, int n_ks)
{
    const long long total = static_cast<long long>(n_mt) * n_ks * PWB_M * PWB_K;
    for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += static_cast<long long>(gridDim.x) * blockDim.x) {
        long long e = i;
        const int ci_l = e % PWB_K; e /= PWB_K;
        const int co_l = e % PWB_M; e /= PWB_M;
        const int ks = e % n_ks; e /= n_ks;
        const int mt = static_cast<int>(e);
        const int co = mt * PWB_M + co_l, ci = ks * PWB_K + ci_l;
        const float v = (co < c_out && ci < c_in) ? w[static_cast<size_t>(co) * c_in + ci] : 0.f;
        const int blk = ci_l >> 4, half = (ci_l >> 3) & 1, c8 = ci_l & 7;
        const size_t step = static_cast<size_t>(mt) * n_ks + ks;
        wp[step * (PWB_A_STEP / 2) + ((static_cast<size_t>(blk) * 2 + half) * PWB_M + co_l) * 8 + c8] = static_cast<unsigned short>(pack_bf16x2(v, 0.f) & 0xffffu);
    }
}

// the operand image: one 256-thread workgroup per (frame tile, K-step, utterance), lane = frame; with LNX the pending LayerNorm is applied
// in fp32 and the result rounded to bf16 (what the reference's LayerNorm module returns under model.to(bfloat16))
template <bool LNX>
__global__ __launch_bounds__(256) void pwb_image_kernel(const bf16_t* __restrict__ x, unsigned char* __restrict__ image, int c_in, int frames,
                                                        int ld, int n_ks, const LnRef ln)
{
    const int nt = blockIdx.x, ks = blockIdx.y, b = blockIdx.z;
    const int lane_t = threadIdx.x;
    const int t = nt * PWB_N + lane_t;
    const bool live = t < frames;
    const bf16_t* __restrict__ xb = x + static_cast<size_t>(b) * c_in * ld + t;
    float mean = 0.f, rstd = 0.f;
    if (LNX && live) {
        const float* st = ln.stats + static_cast<size_t>(b) * 2 * ld;
        mean = st[t]; rstd = st[ld + t];
    }
    float v[PWB_K];
#pragma unroll
    for (int c = 0; c < PWB_K; ++c) {
        const int ci = ks * PWB_K + c;                          // wave-uniform
        v[c] = (live && ci < c_in) ? __uint_as_float(static_cast<unsigned>(xb[static_cast<size_t>(ci) * ld].bits) << 16) : 0.f;
    }
    if (LNX) {
#pragma unroll
        for (int c = 0; c < PWB_K; ++c) {
            const int ci = ks * PWB_K + c;
            if (live && ci < c_in) v[c] = ln_apply(v[c], mean, rstd, ln.gamma[ci], ln.beta[ci]);
        }
    }
    unsigned char* step = image + ((static_cast<size_t>(b) * gridDim.x + nt) * n_ks + ks) * PWB_X_STEP;
#pragma unroll
    for (int bh = 0; bh < 4; ++bh) {                            // (16-channel block, 8-channel half)
        const pwb_u4 row = {pack_bf16x2(v[bh * 8 + 0], v[bh * 8 + 1]), pack_bf16x2(v[bh * 8 + 2], v[bh * 8 + 3]),
                            pack_bf16x2(v[bh * 8 + 4], v[bh * 8 + 5]), pack_bf16x2(v[bh * 8 + 6], v[bh * 8 + 7])};
        *reinterpret_cast<pwb_u4*>(step + (bh * PWB_N + lane_t) * 16) = row;
    }
}

struct PointwiseBf16Args {
    const unsigned char* image; const unsigned char* wp;
    const float* bias; const float* bias2;
    const bf16_t* s0; const bf16_t* s1; const bf16_t* s2;
    void* y;                                 // SWAP: float (frames, batch, c_out); else bf16_t (batch, c_out, ld)
    int c_out, frames, ld_out, n_ks, n_mt, n_nt, batch;
    LnRef ln_s0;
};

__device__ __forceinline__ float pwb_load(const bf16_t* p) { return __uint_as_float(static_cast<unsigned>(p->bits) << 16); }

template <bool SWAP, bool RELU>
__global__ __launch_bounds__(PWB_THREADS, 2) void pwb_gemm_kernel(const PointwiseBf16Args a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const Abuf = smem;                       // [2][PWB_A_STEP]
    unsigned char* const Xbuf = smem + 2 * PWB_A_STEP;      // [2][PWB_X_STEP]

    // XCD-aware, frame-tile-major order: L = (b, nt, mt) -- the workgroups an XCD runs at once are ALL row tiles of a few frame tiles
    const int nwg = gridDim.x, id = blockIdx.x;
    const int xq = nwg >> 3, xr = nwg & 7, xcd = id & 7;
    const int L = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (id >> 3);
    const int ntg = L / a.n_mt;
    const int mt_i = L - ntg * a.n_mt;
    const int b = ntg / a.n_nt;
    const int nt_i = ntg - b * a.n_nt;
    const int m0 = mt_i * PWB_M, n0 = nt_i * PWB_N;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int l15 = lane & 15, kq = lane >> 4;
    const bool wave_active = (m0 + wm * 64) < a.c_out && (n0 + wn * 64) < a.ld_out;

    const unsigned char* __restrict__ wsrc = a.wp + static_cast<size_t>(mt_i) * a.n_ks * PWB_A_STEP;
    const unsigned char* __restrict__ xsrc = a.image + (static_cast<size_t>(b) * a.n_nt + nt_i) * a.n_ks * PWB_X_STEP;

    pwb_f4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

    // one K-step = 24 x 1 KiB LDS-DMA pieces (8 weights + 16 image), 3 per wave
    auto dma_step = [&](int ks, int buf) {
        const unsigned char* ws = wsrc + static_cast<size_t>(ks) * PWB_A_STEP;
        const unsigned char* xs = xsrc + static_cast<size_t>(ks) * PWB_X_STEP;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ws + wave * 1024 + lane * 16),
                                         (__attribute__((address_space(3))) void*)(Abuf + buf * PWB_A_STEP + wave * 1024), 16, 0, 0);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int piece = wave * 2 + j;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xs + piece * 1024 + lane * 16),
                                             (__attribute__((address_space(3))) void*)(Xbuf + buf * PWB_X_STEP + piece * 1024), 16, 0, 0);
        }
    };

    // per-lane fragment bases: k quarter kq -> 16-channel block kq >> 1, 8-channel half kq & 1
    const int a_lane = ((kq * PWB_M) + wm * 64 + l15) * 16;
    const int x_lane = ((kq * PWB_N) + wn * 64 + l15) * 16;

    dma_step(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll 1
    for (int ks = 0; ks < a.n_ks; ++ks) {
        const int buf = ks & 1;
        if (ks + 1 < a.n_ks) dma_step(ks + 1, buf ^ 1);
        if (wave_active) {
            const unsigned char* A = Abuf + buf * PWB_A_STEP + a_lane;
            const unsigned char* X = Xbuf + buf * PWB_X_STEP + x_lane;
            pwb_bf8 bx[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) bx[j] = *reinterpret_cast<const pwb_bf8*>(X + j * 16 * 16);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const pwb_bf8 aw = *reinterpret_cast<const pwb_bf8*>(A + i * 16 * 16);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aw, bx[j], acc[i][j], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    if (!wave_active) return;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + j * 16 + l15;
            const int mb = m0 + wm * 64 + i * 16 + kq * 4;
            if (SWAP) {
                if (n >= a.frames || mb >= a.c_out) continue;
                pwb_f4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = mb + r;
                    float v = 0.f;
                    if (m < a.c_out) {
                        v = acc[i][j][r] + a.bias[m];
                        if (a.bias2) v += a.bias2[m];
                        if (RELU) v = relu_clamp(v);
                    }
                    o[r] = v;
                }
                float* dst = static_cast<float*>(a.y) + (static_cast<size_t>(n) * a.batch + b) * a.c_out + mb;
                if (mb + 3 < a.c_out) *reinterpret_cast<pwb_f4*>(dst) = o;
                else for (int r = 0; r < 4 && mb + r < a.c_out; ++r) dst[r] = o[r];
            } else {
                if (n >= a.ld_out) continue;
                const bool live = n < a.frames;
                float s0m = 0.f, s0r = 0.f;
                if (a.s0 && a.ln_s0.stats) {
                    const float* st = a.ln_s0.stats + static_cast<size_t>(b) * 2 * a.ld_out;
                    s0m = st[n]; s0r = st[a.ld_out + n];
                }
                bf16_t* yb = static_cast<bf16_t*>(a.y);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = mb + r;
                    if (m >= a.c_out) continue;
                    float v = acc[i][j][r] + a.bias[m];
                    if (RELU) v = relu_clamp(v);
                    const size_t off = (static_cast<size_t>(b) * a.c_out + m) * a.ld_out + n;
                    // the skips in python's sum order, in fp32; ONE rounding (as the bf16 node kernels: grouped_conv_impl.h)
                    if (a.s0) v += a.ln_s0.stats ? ln_apply(pwb_load(a.s0 + off), s0m, s0r, a.ln_s0.gamma[m], a.ln_s0.beta[m]) : pwb_load(a.s0 + off);
                    if (a.s1) v += pwb_load(a.s1 + off);
                    if (a.s2) v += pwb_load(a.s2 + off);
                    yb[off].bits = static_cast<unsigned short>(pack_bf16x2(live ? v : 0.f, 0.f) & 0xffffu);
                }
            }
        }
    }
}

static inline int pwb_n_ks(int c_in) { return (c_in + PWB_K - 1) / PWB_K; }
static inline int pwb_n_mt(int c_out) { return (c_out + PWB_M - 1) / PWB_M; }
static inline int pwb_n_nt(int ld) { return (ld + PWB_N - 1) / PWB_N; }

template <bool SWAP, bool RELU>
static int pwb_launch(PointwiseBf16Args a, hipStream_t stream, const char* what)
{
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(pwb_gemm_kernel<SWAP, RELU>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, PWB_LDS);
    if (attr != hipSuccess) {
        set_error("%s: cannot reserve %d bytes of LDS: %s", what, PWB_LDS, hipGetErrorString(attr));
        return static_cast<int>(attr);
    }
    const long long nwg = static_cast<long long>(a.n_mt) * a.n_nt * a.batch;
    NBASR_REQUIRE(nwg < (1ll << 31), NBASR_EINVAL, "%s: too many tiles (%lld)", what, nwg);
    hipLaunchKernelGGL((pwb_gemm_kernel<SWAP, RELU>), dim3(static_cast<unsigned>(nwg)), dim3(PWB_THREADS), PWB_LDS, stream, a);
    return launch_status(what);
}

static int pwb_image(const bf16_t* x, void* ws, int batch, int c_in, int frames, int ld, LnRef ln, hipStream_t stream, const char* what)
{
    const dim3 grid(pwb_n_nt(ld), pwb_n_ks(c_in), batch);
    if (ln.stats) hipLaunchKernelGGL(pwb_image_kernel<true>, grid, dim3(256), 0, stream, x, static_cast<unsigned char*>(ws), c_in, frames, ld, pwb_n_ks(c_in), ln);
    else hipLaunchKernelGGL(pwb_image_kernel<false>, grid, dim3(256), 0, stream, x, static_cast<unsigned char*>(ws), c_in, frames, ld, pwb_n_ks(c_in), ln);
    return launch_status(what);
}

static int pwb_checks(const char* what, const void* x, const void* ws, const void* packed_w, const float* bias, const void* y,
                      int batch, int c_in, int frames, int ld, int c_out)
{
    NBASR_REQUIRE(batch >= 0 && c_in > 0 && c_out > 0 && frames >= 0 && ld >= frames, NBASR_EINVAL, "%s: bad sizes", what);
    if (batch == 0 || frames == 0) return 1;
    NBASR_REQUIRE(batch <= 65535, NBASR_EINVAL, "%s: batch %d > 65535", what, batch);
    NBASR_REQUIRE(x && ws && packed_w && bias && y, NBASR_ENULL, "%s: x, workspace, packed_w, bias, y must be non-NULL", what);
    NBASR_REQUIRE(aligned16(ws) && aligned16(packed_w), NBASR_EALIGN, "%s: workspace and packed weights must be 16-byte aligned", what);
    return NBASR_OK;
}

}  // namespace nbasr

using namespace nbasr;

extern "C" size_t nbasr_pointwise_bf16_weights_bytes(int c_out, int c_in)
{
    if (c_out <= 0 || c_in <= 0) return 0;
    return static_cast<size_t>(pwb_n_mt(c_out)) * pwb_n_ks(c_in) * PWB_A_STEP;
}

extern "C" size_t nbasr_pointwise_bf16_workspace_bytes(int batch, int c_in, int ld)
{
    if (batch <= 0 || c_in <= 0 || ld <= 0) return 0;
    return static_cast<size_t>(batch) * pwb_n_nt(ld) * pwb_n_ks(c_in) * PWB_X_STEP;
}

extern "C" int nbasr_pack_pointwise_weights_bf16(const float* w, void* packed, int c_out, int c_in, nbasr_stream_t stream)
{
    clear_error();
    NBASR_REQUIRE(c_out > 0 && c_in > 0, NBASR_EINVAL, "nbasr_pack_pointwise_weights_bf16: bad sizes");
    NBASR_REQUIRE(w && packed, NBASR_ENULL, "nbasr_pack_pointwise_weights_bf16: NULL pointer");
    NBASR_REQUIRE(aligned16(packed), NBASR_EALIGN, "nbasr_pack_pointwise_weights_bf16: packed buffer must be 16-byte aligned");
    hipLaunchKernelGGL(pwb_pack_weights_kernel, dim3(1024), dim3(256), 0, as_stream(stream), w, static_cast<unsigned short*>(packed), c_out, c_in,
                       pwb_n_mt(c_out), pwb_n_ks(c_in));
    return launch_status("nbasr_pack_pointwise_weights_bf16");
}

extern "C" int nbasr_linear_fused_bf16(const void* x, void* ws, const void* packed_w, const float* bias, const void* skip0,
                                       const void* skip1, const void* skip2, void* y, int batch, int channels_in, int frames, int ld,
                                       int channels_out, const nbasr_deferred_ln* ln, int ln_on_x, int ln_on_skip0, nbasr_stream_t stream)
{
    clear_error();
    const int rc0 = pwb_checks("nbasr_linear_fused_bf16", x, ws, packed_w, bias, y, batch, channels_in, frames, ld, channels_out);
    if (rc0 != NBASR_OK) return rc0 == 1 ? NBASR_OK : rc0;
    NBASR_REQUIRE(ld % 8 == 0, NBASR_EALIGN, "nbasr_linear_fused_bf16: ld=%d must be a multiple of 8 (bf16 rows are pitched to 16 bytes)", ld);
    const bool any_ln = ln && (ln_on_x || (ln_on_skip0 && skip0));
    NBASR_REQUIRE(!any_ln || (ln->stats && ln->gamma && ln->beta), NBASR_ENULL, "nbasr_linear_fused_bf16: deferred LayerNorm needs stats, gamma and beta");
    NBASR_REQUIRE(!(ln && ln_on_x && ln_on_skip0 && skip0) || channels_in == channels_out, NBASR_EINVAL,
                  "nbasr_linear_fused_bf16: one descriptor for x and skip0 needs equal shapes");
    hipStream_t s = as_stream(stream);
    int rc = pwb_image(static_cast<const bf16_t*>(x), ws, batch, channels_in, frames, ld, ln_ref(ln, ln_on_x != 0), s, "nbasr_linear_fused_bf16(image)");
    if (rc != NBASR_OK) return rc;
    PointwiseBf16Args a{};
    a.image = static_cast<const unsigned char*>(ws); a.wp = static_cast<const unsigned char*>(packed_w);
    a.n_mt = pwb_n_mt(channels_out); a.n_ks = pwb_n_ks(channels_in); a.n_nt = pwb_n_nt(ld);
    a.bias = bias; a.s0 = static_cast<const bf16_t*>(skip0); a.s1 = static_cast<const bf16_t*>(skip1); a.s2 = static_cast<const bf16_t*>(skip2);
    a.y = y; a.c_out = channels_out; a.frames = frames; a.ld_out = ld; a.batch = batch;
    a.ln_s0 = ln_ref(ln, ln_on_skip0 != 0 && skip0 != nullptr);
    return pwb_launch<false, true>(a, s, "nbasr_linear_fused_bf16");
}

extern "C" int nbasr_lstm_input_projection_bf16(const void* x, void* ws, const void* packed_w_ih, const float* b_ih, const float* b_hh,
                                                float* gates_ws, int batch, int c_in, int frames, int ld, int hidden,
                                                const nbasr_deferred_ln* ln, nbasr_stream_t stream)
{
    clear_error();
    const int rc0 = pwb_checks("nbasr_lstm_input_projection_bf16", x, ws, packed_w_ih, b_ih, gates_ws, batch, c_in, frames, ld, 4 * hidden);
    if (rc0 != NBASR_OK) return rc0 == 1 ? NBASR_OK : rc0;
    NBASR_REQUIRE(b_hh != nullptr, NBASR_ENULL, "nbasr_lstm_input_projection_bf16: b_hh is NULL");
    NBASR_REQUIRE(hidden % 4 == 0 && aligned16(gates_ws) && ld % 8 == 0, NBASR_EALIGN,
                  "nbasr_lstm_input_projection_bf16: hidden %% 4, ld %% 8 and 16-byte aligned gates_ws required");
    NBASR_REQUIRE(!ln || (ln->stats && ln->gamma && ln->beta), NBASR_ENULL, "nbasr_lstm_input_projection_bf16: deferred LayerNorm needs stats, gamma and beta");
    hipStream_t s = as_stream(stream);
    int rc = pwb_image(static_cast<const bf16_t*>(x), ws, batch, c_in, frames, ld, ln_ref(ln, true), s, "nbasr_lstm_input_projection_bf16(image)");
    if (rc != NBASR_OK) return rc;
    PointwiseBf16Args a{};
    a.image = static_cast<const unsigned char*>(ws); a.wp = static_cast<const unsigned char*>(packed_w_ih);
    a.n_mt = pwb_n_mt(4 * hidden); a.n_ks = pwb_n_ks(c_in); a.n_nt = pwb_n_nt(ld);
    a.bias = b_ih; a.bias2 = b_hh; a.y = gates_ws;
    a.c_out = 4 * hidden; a.frames = frames; a.ld_out = ld; a.batch = batch;
    return pwb_launch<true, false>(a, s, "nbasr_lstm_input_projection_bf16");
}
