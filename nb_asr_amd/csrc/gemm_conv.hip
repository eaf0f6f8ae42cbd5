// Dense (groups = 1) convolution as an implicit GEMM on the fp32 matrix cores of gfx950
// (v_mfma_f32_32x32x2_f32: exact f32 fma chains, so results are f32-for-f32 comparable with the
// reference's oneDNN path), with the PadConvRelu epilogue fused:
//   KW = 8 : the four downsample convs, stride 1 | 2   (reference model.py:82-89, ops.py:24-30)
//   KW = 1 : the `linear` node op                       (reference ops.py:42-50)
//   KW = 1, transposed store: the LSTM input projection x W_ih^T + b_ih + b_hh for all frames at
//            once                                        (reference model.py:100, 118-121)
//
//   D[co][t] = sum_{ci, tap} W[co][ci][tap] * xpad[ci][t * stride + tap - lpad]
//
// GEMM view per utterance: M = c_out, N = output frames, K = c_in * KW.  A workgroup (4 waves, 2x2)
// owns a 128 x 128 tile; a wave owns 64 x 64 as 2 x 2 MFMA 32x32 blocks (64 accumulator VGPRs).
// Per K-step of 64: the weight tile (128 x 64, K-contiguous rows in HBM) is staged k-major in LDS;
// the input is staged as KC = 64/KW channel ROWS of (127*stride + KW) frames -- the sliding window
// is resolved when fragments are read from LDS, so every input element is fetched from global
// memory once per tile and reused KW times from LDS (no im2col, no padded copy).  Zero padding is
// applied by predicate while staging.  Global loads for step s+1 are issued before the MFMAs of
// step s (register staging), so HBM/L2 latency hides under the 8192-cycle MFMA block.  Measured (rocprofv3 PMC,
// round 1): matrix pipe 78.6 % busy at 2.31 GHz = 120 TFLOP/s on full tiles; a bare MFMA loop reaches 150.
//
// Workgroup -> tile order is m-major and XCD-aware: the 8 XCDs each walk a contiguous range of
// tiles, so the 32 CUs of an XCD share one 128-row weight slab in their 4 MiB L2.
// MFMA-bound: 2 * c_out * c_in * KW * frames_out * batch flops per launch.
#include "common.h"

#include <type_traits>

namespace nbasr {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4v __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128;
constexpr int FLUSH = 4;     // K-steps per blocked-summation flush (power of two)
constexpr int LDA = BM;   // lanes run along m for both the staging writes and the fragment reads: conflict-free unpadded

struct GemmConvArgs {
    const float* x; const float* w; const float* bias; const float* bias2;
    const float* s0; const float* s1; const float* s2;
    float* y;
    int c_in, frames_in, ld_in, c_out, frames_out, ld_out, lpad, ktot;
    int n_cover;                 // output columns the tiles must cover (ld_out: pitch columns get zeros)
    int row_stride_t, row_stride_b;   // transposed store: output row of (b, t) = t * row_stride_t + b * row_stride_b
    int n_mt, n_nt, batch;
    LnRef ln_x, ln_s0;           // pending LayerNorm of the input / of skip0 (deferred normalisation, nbasr.h)
};

template <int KW, int STRIDE>
struct Geo {
    static constexpr int BK = (KW == 1) ? 32 : 64;           // K-step (KW = 1 stages 4x more input rows per k: keep registers < 256)
    static constexpr int KC = BK / KW;                       // input channels per K-step
    static constexpr int AREGS = BM * BK / 4 / 256;          // float4 weight chunks per thread per K-step
    static constexpr int XW = (BN - 1) * STRIDE + KW;        // staged frames per channel row
    static constexpr int LDX = XW;                           // lanes run along frames: conflict-free unpadded
    static constexpr int XELEMS = KC * XW;
    static constexpr int XREGS = (XELEMS + 255) / 256;
    static constexpr int LDS_FLOATS = BK * LDA + KC * LDX;
};

// a wave-uniform buffer resource over `bytes` bytes at `p` (both forced into scalar registers: left to itself hipcc treats a resource built
// inside the K loop as divergent and wraps every load in a waterfall loop)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t uniform_rsrc(const float* p, unsigned bytes)
{
    const uintptr_t u = reinterpret_cast<uintptr_t>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(u)), hi = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(u >> 32));
    float* q = reinterpret_cast<float*>((static_cast<uintptr_t>(hi) << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(q, 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

// LNX: the input carries a pending LayerNorm applied while staging (a template flag: its per-item statistics addresses cost the plain
// instances ~36 registers when it was a run-time branch)
template <int KW, int STRIDE, bool SWAP, bool RELU, bool LNX>
__global__ __launch_bounds__(256, 2) void gemm_conv_kernel(const GemmConvArgs a)
{
    using G = Geo<KW, STRIDE>;
    __shared__ float lds[G::LDS_FLOATS];
    constexpr int BK = G::BK;
    float* As = lds;                    // [BK][LDA]  k-major weight tile
    float* Xs = lds + G::BK * LDA;         // [KC][LDX]  input channel rows

    // ---- XCD-aware, m-major tile order (bijective remap, cdna guide T1) ------------------------
    const int nwg = gridDim.x;
    const int id = blockIdx.x;
    const int xq = nwg >> 3, xr = nwg & 7, xcd = id & 7;
    const int L = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (id >> 3);
    const int per_m = a.n_nt * a.batch;
    const int mt_i = L / per_m;
    const int rem = L - mt_i * per_m;
    const int b = rem / a.n_nt;
    const int nt_i = rem - b * a.n_nt;
    const int m0 = mt_i * BM, n0 = nt_i * BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;

    const float* __restrict__ xb = a.x + static_cast<size_t>(b) * a.c_in * a.ld_in;
    const int tin0 = n0 * STRIDE - a.lpad;
    const float* __restrict__ xstats = LNX ? a.ln_x.stats + static_cast<size_t>(b) * 2 * a.ld_in : nullptr;

    // wave-uniform validity of the 32x32 blocks (skip MFMAs on fully out-of-range blocks)
    bool mval[2], nval[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        mval[i] = (m0 + wm * 64 + i * 32) < a.c_out;
        nval[i] = (n0 + wn * 64 + i * 32) < a.n_cover;
    }

    const bool full = mval[0] && mval[1] && nval[0] && nval[1];

    floatx16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- register staging ----------------------------------------------------------------------
    float4 areg[G::AREGS];
    float xreg[G::XREGS];
    const int a_row = tid & 127, a_kh = tid >> 7;   // lane -> weight row; 16-byte k-chunks kh, kh+2, ...

    // Operand fetches are bounds-checked BUFFER loads with one 32-bit offset per item (round 5): the K-step moves the resource's base
    // (scalar arithmetic), the per-item offset is loop-invariant, and what lies outside the tensor -- weight rows beyond c_out, input
    // channels beyond c_in, the zero padding left and right of the utterance -- is an offset beyond num_records, which reads as 0.  Rounds
    // 1-4 kept a 64-bit address and a predicate per item: hoisted out of the K loop they spilled (156-180 bytes of scratch per lane in
    // three of the six instances).
    constexpr unsigned OOB = 0x7ffffff0u;
    const unsigned w_bytes = static_cast<unsigned>(min(static_cast<long long>(a.c_out) * a.ktot * 4, 0x7fffffffll));
    const unsigned x_bytes = static_cast<unsigned>(min(static_cast<long long>(a.c_in) * a.ld_in * 4, 0x7fffffffll));
    const unsigned a_off = (m0 + a_row) < a.c_out ? static_cast<unsigned>(((m0 + a_row) * a.ktot + a_kh * 4) * 4) : OOB;
    unsigned x_off[G::XREGS];
#pragma unroll
    for (int i = 0; i < G::XREGS; ++i) {
        const int e = tid + 256 * i;
        const int ci = e / G::XW;
        const int p = e - ci * G::XW;
        const int t = tin0 + p;
        x_off[i] = (e < G::XELEMS && t >= 0 && t < a.frames_in) ? static_cast<unsigned>((ci * a.ld_in + t) * 4) : OOB;
    }
    auto prefetch = [&](int ks) {
        const int k0 = ks * BK;
        // (k0 floats into the weight rows; the rows' tail beyond ktot -- the last, partial K-step -- must read as zero, not as the next row)
        const __amdgpu_buffer_rsrc_t wr = uniform_rsrc(a.w + k0, w_bytes > static_cast<unsigned>(k0) * 4u ? w_bytes - k0 * 4 : 0);
#pragma unroll
        for (int i = 0; i < G::AREGS; ++i) {
            const bool in_k = k0 + (a_kh + 2 * i) * 4 < a.ktot;
            const floatx4v v = __builtin_bit_cast(floatx4v, __builtin_amdgcn_raw_buffer_load_b128(wr, in_k ? a_off + i * 32 : OOB, 0, 0));
            areg[i] = make_float4(v[0], v[1], v[2], v[3]);
        }
        const int ci0 = ks * G::KC;
        [[maybe_unused]] const __amdgpu_buffer_rsrc_t sr = uniform_rsrc(xstats, LNX ? static_cast<unsigned>(2 * a.ld_in * 4) : 0u);
        const unsigned ch_left = LNX && ci0 < a.c_in ? static_cast<unsigned>((a.c_in - ci0) * 4) : 0u;
        [[maybe_unused]] const __amdgpu_buffer_rsrc_t gr = uniform_rsrc(LNX ? a.ln_x.gamma + ci0 : nullptr, ch_left);
        [[maybe_unused]] const __amdgpu_buffer_rsrc_t br = uniform_rsrc(LNX ? a.ln_x.beta + ci0 : nullptr, ch_left);
        const unsigned row0_bytes = static_cast<unsigned>(ci0) * a.ld_in * 4;
        const __amdgpu_buffer_rsrc_t xr = uniform_rsrc(xb + static_cast<size_t>(ci0) * a.ld_in, x_bytes > row0_bytes ? x_bytes - row0_bytes : 0);
#pragma unroll
        for (int i = 0; i < G::XREGS; ++i) {
            float v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, x_off[i], 0, 0));
            if constexpr (LNX) {
                const int e = tid + 256 * i;
                const int ci = ci0 + e / G::XW, t = tin0 + e % G::XW;
                if (x_off[i] != OOB && ci < a.c_in) {
                    // (statistics through a resource + 32-bit offsets too: two 64-bit addresses per item were the LayerNorm instances' spills)
                    const float mean = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(sr, t * 4, 0, 0));
                    const float rstd = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(sr, (a.ld_in + t) * 4, 0, 0));
                    const float gam = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(gr, (e / G::XW) * 4, 0, 0));
                    const float bet = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(br, (e / G::XW) * 4, 0, 0));
                    v = ln_apply(v, mean, rstd, gam, bet);
                }
            }
            xreg[i] = v;
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < G::AREGS; ++i) {
            float* dst = As + ((a_kh + 2 * i) * 4) * LDA + a_row;
            dst[0 * LDA] = areg[i].x; dst[1 * LDA] = areg[i].y; dst[2 * LDA] = areg[i].z; dst[3 * LDA] = areg[i].w;
        }
#pragma unroll
        for (int i = 0; i < G::XREGS; ++i) {
            const int e = tid + 256 * i;
            const int ci = e / G::XW;
            const int p = e - ci * G::XW;
            if (e < G::XELEMS) Xs[ci * G::LDX + p] = xreg[i];
        }
    };

    const int nks = (a.ktot + BK - 1) / BK;
    const float* a_base = As + half * LDA + wm * 64 + l31;
    const float* x_base = (KW == 1) ? Xs + half * G::LDX + (wn * 64 + l31)
                                    : Xs + (wn * 64 + l31) * STRIDE + half;

    floatx16 part[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) part[i][j][r] = 0.f;

    prefetch(0);
    for (int ks = 0; ks < nks; ++ks) {
        __syncthreads();                 // previous step's fragment reads are done
        commit();
        __syncthreads();
        if (ks + 1 < nks) prefetch(ks + 1);

        auto mma_step = [&](auto all_valid) {
#pragma unroll
            for (int kk = 0; kk < BK / 2; ++kk) {
                const int xo = (KW == 1) ? 2 * kk * G::LDX : (kk / (KW / 2 > 0 ? KW / 2 : 1)) * G::LDX + 2 * (kk % (KW / 2 > 0 ? KW / 2 : 1));
                float av[2], bv[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) av[i] = a_base[2 * kk * LDA + i * 32];
#pragma unroll
                for (int j = 0; j < 2; ++j) bv[j] = x_base[xo + j * 32 * STRIDE];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        if (decltype(all_valid)::value || (mval[i] && nval[j]))
                            part[i][j] = SWAP ? __builtin_amdgcn_mfma_f32_32x32x2f32(bv[j], av[i], part[i][j], 0, 0, 0)
                                              : __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], part[i][j], 0, 0, 0);
            }
        };
        if (full) mma_step(std::true_type{}); else mma_step(std::false_type{});
        // blocked summation: partial tiles are flushed into the running total every FLUSH K-steps, so rounding
        // error grows like sqrt(FLUSH * BK) + sqrt(K / (FLUSH * BK)) instead of sqrt(K); flushing drains the MFMA
        // pipeline, hence not every step
        if ((ks & (FLUSH - 1)) == FLUSH - 1 || ks + 1 == nks) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] += part[i][j];
#pragma unroll
                    for (int r = 0; r < 16; ++r) part[i][j][r] = 0.f;
                }
        }
    }

    // ---- epilogue ------------------------------------------------------------------------------
    // C/D layout of the 32x32 MFMA: column = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (!(mval[i] && nval[j])) continue;
            const int mb = m0 + wm * 64 + i * 32;     // c_out block base
            const int nb = n0 + wn * 64 + j * 32;     // frame block base
            if (!SWAP) {
                const int n = nb + l31;
                if (n >= a.ld_out) continue;
                const bool live = n < a.frames_out;
                float s0m = 0.f, s0r = 0.f;
                if (a.s0 && a.ln_s0.stats) {
                    const float* st = a.ln_s0.stats + static_cast<size_t>(b) * 2 * a.ld_out;
                    s0m = st[n]; s0r = st[a.ld_out + n];
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mb + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (m >= a.c_out) continue;
                    float v = acc[i][j][r] + a.bias[m];
                    if (RELU) v = relu_clamp(v);
                    const size_t off = (static_cast<size_t>(b) * a.c_out + m) * a.ld_out + n;
                    if (a.s0) v += a.ln_s0.stats ? ln_apply(a.s0[off], s0m, s0r, a.ln_s0.gamma[m], a.ln_s0.beta[m]) : a.s0[off];
                    if (a.s1) v += a.s1[off];
                    if (a.s2) v += a.s2[off];
                    a.y[off] = live ? v : 0.f;
                }
            } else {
                // transposed store: y[row(b, t) * c_out + m], lanes along m
                const int m = mb + l31;
                if (m >= a.c_out) continue;
                float bsum = a.bias[m];
                if (a.bias2) bsum += a.bias2[m];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int t = nb + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (t >= a.frames_out) continue;
                    float v = acc[i][j][r] + bsum;
                    if (RELU) v = relu_clamp(v);
                    a.y[(static_cast<size_t>(t) * a.row_stride_t + static_cast<size_t>(b) * a.row_stride_b) * a.c_out + m] = v;
                }
            }
        }
    }
}

template <int KW, int STRIDE, bool SWAP, bool RELU>
static int launch_gemm_conv(GemmConvArgs a, hipStream_t stream, const char* what)
{
    a.n_mt = (a.c_out + BM - 1) / BM;
    a.n_cover = SWAP ? a.frames_out : a.ld_out;
    a.n_nt = (a.n_cover + BN - 1) / BN;
    const long long nwg = static_cast<long long>(a.n_mt) * a.n_nt * a.batch;
    if (nwg == 0) return NBASR_OK;
    NBASR_REQUIRE(nwg < (1ll << 31), NBASR_EINVAL, "%s: too many tiles (%lld)", what, nwg);
    if (a.ln_x.stats) hipLaunchKernelGGL((gemm_conv_kernel<KW, STRIDE, SWAP, RELU, true>), dim3(static_cast<unsigned>(nwg)), dim3(256), 0, stream, a);
    else              hipLaunchKernelGGL((gemm_conv_kernel<KW, STRIDE, SWAP, RELU, false>), dim3(static_cast<unsigned>(nwg)), dim3(256), 0, stream, a);
    return launch_status(what);
}

// internal entry used by lstm.hip: gates(t, b, 4H) = x(b, :, t) . w_ih^T + b_ih + b_hh  (TIME-major, so that the
// recurrence reads one contiguous batch x 4H slab per step)
int lstm_input_projection(const float* x, const float* w_ih, const float* b_ih, const float* b_hh, float* gates,
                          int batch, int c_in, int frames, int ld, int rows4h, LnRef ln, hipStream_t stream)
{
    GemmConvArgs a{};
    a.x = x; a.w = w_ih; a.bias = b_ih; a.bias2 = b_hh; a.y = gates;
    a.c_in = c_in; a.frames_in = frames; a.ld_in = ld; a.c_out = rows4h; a.frames_out = frames; a.ld_out = rows4h;
    a.lpad = 0; a.ktot = c_in; a.batch = batch;
    a.row_stride_t = batch; a.row_stride_b = 1;
    a.ln_x = ln;
    return launch_gemm_conv<1, 1, true, false>(a, stream, "nbasr_lstm_input_projection");
}

}  // namespace nbasr

using namespace nbasr;

extern "C" int nbasr_dense_conv1d_fused(const float* x, const float* w, const float* bias, const float* skip0,
                                        const float* skip1, const float* skip2, float* y, int batch, int c_in,
                                        int frames_in, int ld_in, int c_out, int ld_out, int kernel, int stride,
                                        const nbasr_deferred_ln* ln, int ln_on_x, int ln_on_skip0, nbasr_stream_t stream)
{
    clear_error();
    NBASR_REQUIRE(batch >= 0 && c_in > 0 && c_out > 0 && frames_in >= 0, NBASR_EINVAL, "nbasr_dense_conv1d_fused: bad sizes");
    NBASR_REQUIRE((kernel == 8 && (stride == 1 || stride == 2)) || (kernel == 1 && stride == 1), NBASR_EINVAL,
                  "nbasr_dense_conv1d_fused: (kernel=%d, stride=%d) unsupported; model uses k=8 s in {1,2} and k=1 s=1", kernel, stride);
    const int frames_out = (frames_in + stride - 1) / stride;
    NBASR_REQUIRE(ld_in >= frames_in, NBASR_EINVAL, "nbasr_dense_conv1d_fused: ld_in=%d < frames_in=%d", ld_in, frames_in);
    NBASR_REQUIRE(ld_out >= frames_out && ld_out % 4 == 0, NBASR_EALIGN,
                  "nbasr_dense_conv1d_fused: ld_out=%d must be >= %d output frames and a multiple of 4", ld_out, frames_out);
    NBASR_REQUIRE((c_in * kernel) % 4 == 0 && aligned16(w), NBASR_EALIGN,
                  "nbasr_dense_conv1d_fused: c_in*kernel=%d must be a multiple of 4 and w 16-byte aligned", c_in * kernel);
    if (batch == 0 || frames_out == 0) return NBASR_OK;
    NBASR_REQUIRE(x && w && bias && y, NBASR_ENULL, "nbasr_dense_conv1d_fused: x, w, bias, y must be non-NULL");
    const bool any_ln = ln && (ln_on_x || (ln_on_skip0 && skip0));
    NBASR_REQUIRE(!any_ln || (ln->stats && ln->gamma && ln->beta), NBASR_ENULL,
                  "nbasr_dense_conv1d_fused: deferred LayerNorm needs stats, gamma and beta");
    NBASR_REQUIRE(!(ln && ln_on_x && ln_on_skip0 && skip0) || (c_in == c_out && ld_in == ld_out), NBASR_EINVAL,
                  "nbasr_dense_conv1d_fused: one descriptor for x and skip0 needs equal shapes");
    GemmConvArgs a{};
    a.x = x; a.w = w; a.bias = bias; a.bias2 = nullptr; a.s0 = skip0; a.s1 = skip1; a.s2 = skip2; a.y = y;
    a.c_in = c_in; a.frames_in = frames_in; a.ld_in = ld_in; a.c_out = c_out; a.frames_out = frames_out; a.ld_out = ld_out;
    a.lpad = pad_left(kernel, 1, stride); a.ktot = c_in * kernel; a.batch = batch;
    a.ln_x = ln_ref(ln, ln_on_x != 0); a.ln_s0 = ln_ref(ln, ln_on_skip0 != 0 && skip0 != nullptr);
    hipStream_t s = as_stream(stream);
    if (kernel == 8 && stride == 1) return launch_gemm_conv<8, 1, false, true>(a, s, "nbasr_dense_conv1d_fused");
    if (kernel == 8 && stride == 2) return launch_gemm_conv<8, 2, false, true>(a, s, "nbasr_dense_conv1d_fused");
    return launch_gemm_conv<1, 1, false, true>(a, s, "nbasr_dense_conv1d_fused");
}

// y(batch, c_out, ld_out) = w(c_out, c_in) . x(batch, c_in, ld_in) + bias, no activation: the plain per-frame linear map
// (the front-end's DFT and mel filterbank, frontend.hip); pitch columns frames..ld_out-1 are written as zero
extern "C" int nbasr_pointwise_linear(const float* x, const float* w, const float* bias, float* y, int batch, int c_in,
                                      int frames, int ld_in, int c_out, int ld_out, nbasr_stream_t stream)
{
    clear_error();
    NBASR_REQUIRE(batch >= 0 && c_in > 0 && c_out > 0 && frames >= 0, NBASR_EINVAL, "nbasr_pointwise_linear: bad sizes");
    NBASR_REQUIRE(ld_in >= frames, NBASR_EINVAL, "nbasr_pointwise_linear: ld_in=%d < frames=%d", ld_in, frames);
    NBASR_REQUIRE(ld_out >= frames && ld_out % 4 == 0, NBASR_EALIGN,
                  "nbasr_pointwise_linear: ld_out=%d must be >= %d frames and a multiple of 4", ld_out, frames);
    NBASR_REQUIRE(c_in % 4 == 0 && aligned16(w), NBASR_EALIGN, "nbasr_pointwise_linear: c_in=%d must be a multiple of 4 and w 16-byte aligned", c_in);
    if (batch == 0 || frames == 0) return NBASR_OK;
    NBASR_REQUIRE(x && w && bias && y, NBASR_ENULL, "nbasr_pointwise_linear: x, w, bias, y must be non-NULL");
    GemmConvArgs a{};
    a.x = x; a.w = w; a.bias = bias; a.y = y;
    a.c_in = c_in; a.frames_in = frames; a.ld_in = ld_in; a.c_out = c_out; a.frames_out = frames; a.ld_out = ld_out;
    a.lpad = 0; a.ktot = c_in; a.batch = batch;
    return launch_gemm_conv<1, 1, false, false>(a, as_stream(stream), "nbasr_pointwise_linear");
}

// y(batch, c_out, ld_out)[t] = sum_{ci,j} w[co][ci][j] * xpad(batch, c_in, .)[t + j - lpad] + bias, k = 8 taps, stride 1, NO activation and a
// caller-chosen left padding: the input gradient of a downsample conv is this with the flipped, channel-transposed kernel over the
// zero-stuffed output gradient (autograd.py)
extern "C" int nbasr_dense_conv1d_linear(const float* x, const float* w, const float* bias, float* y, int batch, int c_in, int frames,
                                         int ld_in, int c_out, int ld_out, int kernel, int lpad, nbasr_stream_t stream)
{
    clear_error();
    NBASR_REQUIRE(batch >= 0 && c_in > 0 && c_out > 0 && frames >= 0, NBASR_EINVAL, "nbasr_dense_conv1d_linear: bad sizes");
    NBASR_REQUIRE(kernel == 8 && lpad >= 0 && lpad < kernel, NBASR_EINVAL, "nbasr_dense_conv1d_linear: kernel=%d (8), lpad=%d (0..7)", kernel, lpad);
    NBASR_REQUIRE(ld_in >= frames, NBASR_EINVAL, "nbasr_dense_conv1d_linear: ld_in=%d < frames=%d", ld_in, frames);
    NBASR_REQUIRE(ld_out >= frames && ld_out % 4 == 0, NBASR_EALIGN, "nbasr_dense_conv1d_linear: ld_out=%d must be >= %d frames and a multiple of 4", ld_out, frames);
    NBASR_REQUIRE((c_in * kernel) % 4 == 0 && aligned16(w), NBASR_EALIGN, "nbasr_dense_conv1d_linear: c_in*kernel must be a multiple of 4 and w 16-byte aligned");
    if (batch == 0 || frames == 0) return NBASR_OK;
    NBASR_REQUIRE(x && w && bias && y, NBASR_ENULL, "nbasr_dense_conv1d_linear: x, w, bias, y must be non-NULL");
    GemmConvArgs a{};
    a.x = x; a.w = w; a.bias = bias; a.y = y;
    a.c_in = c_in; a.frames_in = frames; a.ld_in = ld_in; a.c_out = c_out; a.frames_out = frames; a.ld_out = ld_out;
    a.lpad = lpad; a.ktot = c_in * kernel; a.batch = batch;
    return launch_gemm_conv<8, 1, false, false>(a, as_stream(stream), "nbasr_dense_conv1d_linear");
}
