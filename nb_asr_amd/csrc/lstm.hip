// nn.LSTM(1200 -> 500, batch_first, one layer, zero initial state) and the CTC head nn.Linear
// (reference model.py:100-103, forward model.py:118-124), gate order i, f, g, o, bias b_ih + b_hh.
//
//  1. input projection for ALL frames at once: one fp32-MFMA GEMM (gemm_conv.hip, transposed store)
//     gates[t][b][4H] = x[b][:, t] . W_ih^T + b_ih + b_hh      -- the reference's permute(0,2,1) is
//     folded into the operand loader, nothing is transposed in memory;
//  2. the recurrence: one launch per frame.  A workgroup owns 16 hidden units x 16 utterances and
//     all four gates; its 8 waves split K = hidden eight ways (v_mfma_f32_16x16x4_f32, 16-byte
//     operand loads: the 4 components of a lane's float4 feed 4 consecutive MFMAs, which is a
//     k-permutation applied identically to W_hh and h and therefore leaves the sums unchanged),
//     partial tiles are reduced through LDS and the gate nonlinearities, cell update and h store are
//     fused behind the reduction.  h_{t-1} is read straight from the (batch, frames, hidden) output.
//  3. head: logits = h . W^T + b on 16x16x4 MFMA tiles (classes padded to 64 in registers only).
//
// The recurrence is latency-bound (frames dependent steps); steps are plain stream-ordered launches.
#include "common.h"

namespace nbasr {

typedef float floatx4 __attribute__((ext_vector_type(4)));

int lstm_input_projection(const float* x, const float* w_ih, const float* b_ih, const float* b_hh, float* gates,
                          int batch, int c_in, int frames, int ld, int rows4h, LnRef ln, hipStream_t stream);

__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }

// grid: (ceil(hidden/16), ceil(batch/16)); block 512 = 8 waves, each owning one eighth of K per round
constexpr int LSTM_WAVES = 8;
constexpr int LSTM_CHUNKS = 4;      // 16-deep k chunks per wave per round (8 waves x 4 x 16 = 512 >= hidden 500)

__global__ __launch_bounds__(64 * LSTM_WAVES) void lstm_step_kernel(
    const float* __restrict__ gates_in,   // (frames, batch, 4*hidden): input projection incl. biases, time-major
    const float* __restrict__ w_hh,       // (4*hidden, hidden)
    float* __restrict__ cell,             // (batch, hidden) running cell state
    float* h_out,                         // (batch, frames, hidden); row t-1 is read, row t written
    int batch, int frames, int hidden, int t)
{
    __shared__ float red[LSTM_WAVES][4][4][64];    // [wave][gate][reg][lane]

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i16 = lane & 15, kq = lane >> 4;
    const int j0 = blockIdx.x * 16, b0 = blockIdx.y * 16;

    // epilogue role of threads 0..255: one (hidden unit, utterance) each, lanes along the HIDDEN index so the gate
    // reads, the cell state and the h store are 64-byte contiguous per 16 lanes.  Their gate pre-activations come
    // from HBM (written once by the input GEMM): issue those loads FIRST so their latency hides under the matmul.
    const int jj = threadIdx.x & 15, bb = (threadIdx.x >> 4) & 15;
    const int ej = j0 + jj, eb = b0 + bb;
    const bool e_ok = threadIdx.x < 256 && ej < hidden && eb < batch;
    float pre[4] = {0.f, 0.f, 0.f, 0.f};
    float c_prev = 0.f;
    if (e_ok) {
        const float* gin = gates_in + (static_cast<size_t>(t) * batch + eb) * (4 * hidden);
#pragma unroll
        for (int g = 0; g < 4; ++g) pre[g] = gin[g * hidden + ej];
        if (t > 0) c_prev = cell[static_cast<size_t>(eb) * hidden + ej];
    }

    floatx4 acc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = floatx4{0.f, 0.f, 0.f, 0.f};

    if (t > 0) {
        const int kchunks = (hidden + 15) / 16;              // 16 k per chunk
        const bool row_ok = (j0 + i16) < hidden;
        const bool col_ok = (b0 + i16) < batch;
        const float* hrow = h_out + (static_cast<size_t>(b0 + i16) * frames + (t - 1)) * hidden;
        // wave w owns chunks [4w + 32r, 4w + 32r + 4) of round r; all 20 16-byte loads of a round are issued before
        // the first MFMA so their L2 latencies overlap (one round covers hidden <= 512)
        for (int base = wave * LSTM_CHUNKS; base < kchunks; base += LSTM_WAVES * LSTM_CHUNKS) {
            float4 hv[LSTM_CHUNKS], wv[LSTM_CHUNKS][4];
#pragma unroll
            for (int c = 0; c < LSTM_CHUNKS; ++c) {
                const int k = (base + c) * 16 + kq * 4;
                const bool kok = k < hidden;                  // hidden % 4 == 0: whole float4 in or out
                hv[c] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (col_ok && kok) hv[c] = *reinterpret_cast<const float4*>(hrow + k);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    wv[c][g] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (row_ok && kok)
                        wv[c][g] = *reinterpret_cast<const float4*>(w_hh + (static_cast<size_t>(g) * hidden + j0 + i16) * hidden + k);
                }
            }
            // consecutive MFMAs go to different accumulators (40-cycle dependent latency vs 32-cycle issue)
#pragma unroll
            for (int c = 0; c < LSTM_CHUNKS; ++c) {
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[c][g].x, hv[c].x, acc[g], 0, 0, 0);
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[c][g].y, hv[c].y, acc[g], 0, 0, 0);
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[c][g].z, hv[c].z, acc[g], 0, 0, 0);
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[c][g].w, hv[c].w, acc[g], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][g][r][lane] = acc[g][r];
    __syncthreads();

    // C/D layout of the tiles: col = lane & 15 (utterance), row = (lane >> 4) * 4 + reg (hidden unit)
    //   =>  element (jj, bb) sits at reg = jj & 3, lane = (jj >> 2) * 16 + bb
    if (!e_ok) return;
    const int pr = jj & 3, pl = (jj >> 2) * 16 + bb;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        float s = red[0][g][pr][pl];
#pragma unroll
        for (int w = 1; w < LSTM_WAVES; ++w) s += red[w][g][pr][pl];
        pre[g] += s;
    }
    const float c_new = sigmoidf_(pre[1]) * c_prev + sigmoidf_(pre[0]) * tanhf(pre[2]);
    const float h_new = sigmoidf_(pre[3]) * tanhf(c_new);
    cell[static_cast<size_t>(eb) * hidden + ej] = c_new;
    h_out[(static_cast<size_t>(eb) * frames + t) * hidden + ej] = h_new;
}

// ---- the recurrence on a fragment-ordered copy of w_hh -------------------------------------------------------------------
// Stamps on the kernel above: the 160 operand wave-loads of a workgroup (each touching 16 rows x 64 B of the (4H, H) matrix)
// need 4 100 - 8 000 cycles to land -- the L2 -> CU fill rate (~20 B/clk per CU), not latency -- while half of the CUs have
// no workgroup at all.  So (1) the weight is re-laid-out ONCE so that every wave-load is 1 KiB contiguous, and (2) a
// workgroup owns 8 hidden units instead of 16 (252 workgroups at H = 500, B = 64: 96 KiB of operands each instead of 160).
// MFMA tile rows are (hidden unit, gate) = (row >> 2, row & 3), so the four gates of a (unit, utterance) pair sit in the four
// accumulator registers of ONE lane (C/D row = (lane >> 4) * 4 + reg): after the cross-wave K reduction the cell update
// needs no further exchange.  Same fmaf chains per output as the kernel above up to the order of the 8 wave partials.
constexpr int LSTMP_HU = 8;                                   // hidden units per workgroup (2 MFMA row tiles of 4 units x 4 gates)

// packed[((slice * kchunks_p + kc) * 2 + mt) * 64 + lane] (float4) = w_hh[gate*H + unit][kc*16 + (lane >> 4)*4 .. +3] with
// unit = slice*8 + mt*4 + ((lane & 15) >> 2), gate = lane & 3; zero outside the matrix; kchunks_p: chunks rounded up to 4
__global__ __launch_bounds__(256) void lstm_pack_whh_kernel(const float* __restrict__ w_hh, float4* __restrict__ packed, int hidden,
                                                            int slices, int kchunks_p)
{
    const size_t total = static_cast<size_t>(slices) * kchunks_p * 2 * 64;
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += static_cast<size_t>(gridDim.x) * blockDim.x) {
        size_t e = i;
        const int lane = e % 64; e /= 64;
        const int mt = e % 2; e /= 2;
        const int kc = e % kchunks_p; e /= kchunks_p;
        const int slice = static_cast<int>(e);
        const int unit = slice * LSTMP_HU + mt * 4 + ((lane & 15) >> 2), gate = lane & 3, k = kc * 16 + (lane >> 4) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (unit < hidden && k < hidden) v = *reinterpret_cast<const float4*>(w_hh + (static_cast<size_t>(gate) * hidden + unit) * hidden + k);
        packed[i] = v;
    }
}

__global__ __launch_bounds__(64 * LSTM_WAVES) void lstm_step_packed_kernel(
    const float* __restrict__ gates_in,   // (frames, batch, 4*hidden)
    const float4* __restrict__ wp,        // packed w_hh
    float* __restrict__ cell, float* h_out, int batch, int frames, int hidden, int kchunks_p, int t)
{
    __shared__ float red[LSTM_WAVES][2][4][64];    // [wave][row tile][reg = gate][lane]

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i16 = lane & 15, kq = lane >> 4;
    const int j0 = blockIdx.x * LSTMP_HU, b0 = blockIdx.y * 16;

    // epilogue role of threads 0..127: one (hidden unit, utterance) each; gate pre-activations issued first (HBM / L2)
    const int jj = threadIdx.x & 7, bb = (threadIdx.x >> 3) & 15;
    const int ej = j0 + jj, eb = b0 + bb;
    const bool e_ok = threadIdx.x < 128 && ej < hidden && eb < batch;
    float pre[4] = {0.f, 0.f, 0.f, 0.f};
    float c_prev = 0.f;
    if (e_ok) {
        const float* gin = gates_in + (static_cast<size_t>(t) * batch + eb) * (4 * hidden);
#pragma unroll
        for (int g = 0; g < 4; ++g) pre[g] = gin[g * hidden + ej];
        if (t > 0) c_prev = cell[static_cast<size_t>(eb) * hidden + ej];
    }

    floatx4 acc[2] = {floatx4{0.f, 0.f, 0.f, 0.f}, floatx4{0.f, 0.f, 0.f, 0.f}};
    if (t > 0) {
        const bool col_ok = (b0 + i16) < batch;
        const float* hrow = h_out + (static_cast<size_t>(b0 + i16) * frames + (t - 1)) * hidden;
        const float4* wslice = wp + static_cast<size_t>(blockIdx.x) * kchunks_p * 2 * 64 + lane;
        for (int base = wave * LSTM_CHUNKS; base < kchunks_p; base += LSTM_WAVES * LSTM_CHUNKS) {
            float4 hv[LSTM_CHUNKS], wv[LSTM_CHUNKS][2];
#pragma unroll
            for (int c = 0; c < LSTM_CHUNKS; ++c) {
                const int k = (base + c) * 16 + kq * 4;
                hv[c] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (col_ok && k < hidden) hv[c] = *reinterpret_cast<const float4*>(hrow + k);
                wv[c][0] = wslice[static_cast<size_t>(base + c) * 2 * 64];
                wv[c][1] = wslice[static_cast<size_t>(base + c) * 2 * 64 + 64];
            }
#pragma unroll
            for (int c = 0; c < LSTM_CHUNKS; ++c) {
#pragma unroll
                for (int m = 0; m < 2; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[c][m].x, hv[c].x, acc[m], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < 2; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[c][m].y, hv[c].y, acc[m], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < 2; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[c][m].z, hv[c].z, acc[m], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < 2; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[c][m].w, hv[c].w, acc[m], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][m][r][lane] = acc[m][r];
    __syncthreads();

    // tile row = unit_in_tile * 4 + gate = (lane >> 4) * 4 + reg, column = utterance = lane & 15
    //   => unit jj of the slice: tile jj >> 2, lane = (jj & 3) * 16 + bb, reg = gate
    if (!e_ok) return;
    const int mt = jj >> 2, pl = (jj & 3) * 16 + bb;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        float s = red[0][mt][g][pl];
#pragma unroll
        for (int w = 1; w < LSTM_WAVES; ++w) s += red[w][mt][g][pl];
        pre[g] += s;
    }
    const float c_new = sigmoidf_(pre[1]) * c_prev + sigmoidf_(pre[0]) * tanhf(pre[2]);
    const float h_new = sigmoidf_(pre[3]) * tanhf(c_new);
    cell[static_cast<size_t>(eb) * hidden + ej] = c_new;
    h_out[(static_cast<size_t>(eb) * frames + t) * hidden + ej] = h_new;
}

// logits(rows, classes) = h(rows, features) . w(classes, features)^T + bias; classes <= 64.
// BCT = true: h is the encoder output (batch, features, ld) and row = (b, t)  (use_rnn=False model).
template <bool BCT>
__global__ __launch_bounds__(256) void head_kernel(
    const float* __restrict__ h, const float* __restrict__ w, const float* __restrict__ bias,
    float* __restrict__ out, int rows, int features, int classes, int frames, int ld, const LnRef ln)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i16 = lane & 15, kq = lane >> 4;
    const int r0 = (blockIdx.x * 4 + wave) * 16;
    if (r0 >= rows) return;

    // blocked accumulation: 128 products go into `part`, which is then folded into `acc` -- the rounding error of a 1 200-term dot
    // product then grows like that of the blocked / vectorised sums of a CPU GEMM instead of one long sequential chain
    floatx4 acc[4], part[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) { acc[n] = floatx4{0.f, 0.f, 0.f, 0.f}; part[n] = floatx4{0.f, 0.f, 0.f, 0.f}; }

    const int row = r0 + i16;
    const bool row_ok = row < rows;
    const int kchunks = (features + 15) / 16;
    for (int c = 0; c < kchunks; ++c) {
        if ((c & 7) == 0 && c) {
#pragma unroll
            for (int n = 0; n < 4; ++n) { acc[n] += part[n]; part[n] = floatx4{0.f, 0.f, 0.f, 0.f}; }
        }
        const int k = c * 16 + kq * 4;
        const bool kok = k < features;                        // features % 4 == 0
        float av[4] = {0.f, 0.f, 0.f, 0.f};
        if (row_ok && kok) {
            if (!BCT) {
                const float4 v = *reinterpret_cast<const float4*>(h + static_cast<size_t>(row) * features + k);
                av[0] = v.x; av[1] = v.y; av[2] = v.z; av[3] = v.w;
            } else {
                const int bb = row / frames, tt = row - bb * frames;
                const float* p = h + (static_cast<size_t>(bb) * features + k) * ld + tt;
#pragma unroll
                for (int e = 0; e < 4; ++e) av[e] = p[static_cast<size_t>(e) * ld];
                if (ln.stats) {
                    const float* st = ln.stats + static_cast<size_t>(bb) * 2 * ld;
                    const float mean = st[tt], rstd = st[ld + tt];
#pragma unroll
                    for (int e = 0; e < 4; ++e) av[e] = ln_apply(av[e], mean, rstd, ln.gamma[k + e], ln.beta[k + e]);
                }
            }
        }
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int cls = n * 16 + i16;
            float4 wv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (cls < classes && kok) wv = *reinterpret_cast<const float4*>(w + static_cast<size_t>(cls) * features + k);
            part[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0], wv.x, part[n], 0, 0, 0);
            part[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1], wv.y, part[n], 0, 0, 0);
            part[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[2], wv.z, part[n], 0, 0, 0);
            part[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[3], wv.w, part[n], 0, 0, 0);
        }
    }
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[n] += part[n];
    // D[m = row][n = class]: col = lane & 15 (class), row = (lane >> 4) * 4 + reg
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int cls = n * 16 + i16;
        if (cls >= classes) continue;
        const float bv = bias[cls];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rr = r0 + kq * 4 + r;
            if (rr < rows) out[static_cast<size_t>(rr) * classes + cls] = acc[n][r] + bv;
        }
    }
}

}  // namespace nbasr

using namespace nbasr;

static int lstm_check(const char* what, int batch, int c_in, int frames, int ld, int hidden)
{
    NBASR_REQUIRE(batch >= 0 && c_in > 0 && frames >= 0 && hidden > 0 && ld >= frames, NBASR_EINVAL, "%s: bad sizes", what);
    NBASR_REQUIRE(hidden % 4 == 0 && c_in % 4 == 0, NBASR_EALIGN, "%s: hidden=%d and c_in=%d must be multiples of 4", what, hidden, c_in);
    return NBASR_OK;
}

extern "C" int nbasr_lstm_input_projection(const float* x, const float* w_ih, const float* b_ih, const float* b_hh,
                                           float* gates_ws, int batch, int c_in, int frames, int ld, int hidden,
                                           const nbasr_deferred_ln* ln, nbasr_stream_t stream)
{
    clear_error();
    const int rc = lstm_check("nbasr_lstm_input_projection", batch, c_in, frames, ld, hidden);
    if (rc != NBASR_OK) return rc;
    if (batch == 0 || frames == 0) return NBASR_OK;
    NBASR_REQUIRE(x && w_ih && b_ih && b_hh && gates_ws, NBASR_ENULL, "nbasr_lstm_input_projection: NULL pointer");
    NBASR_REQUIRE(aligned16(w_ih), NBASR_EALIGN, "nbasr_lstm_input_projection: w_ih must be 16-byte aligned");
    NBASR_REQUIRE(!ln || (ln->stats && ln->gamma && ln->beta), NBASR_ENULL, "nbasr_lstm_input_projection: deferred LayerNorm needs stats, gamma and beta");
    return lstm_input_projection(x, w_ih, b_ih, b_hh, gates_ws, batch, c_in, frames, ld, 4 * hidden, ln_ref(ln, true), as_stream(stream));
}

extern "C" int nbasr_lstm_recurrence(const float* gates_ws, const float* w_hh, float* cell_ws, float* h_out, int batch,
                                     int frames, int hidden, nbasr_stream_t stream)
{
    clear_error();
    const int rc = lstm_check("nbasr_lstm_recurrence", batch, 4, frames, frames, hidden);
    if (rc != NBASR_OK) return rc;
    if (batch == 0 || frames == 0) return NBASR_OK;
    NBASR_REQUIRE(gates_ws && w_hh && cell_ws && h_out, NBASR_ENULL, "nbasr_lstm_recurrence: NULL pointer");
    NBASR_REQUIRE(aligned16(w_hh) && aligned16(h_out), NBASR_EALIGN, "nbasr_lstm_recurrence: w_hh, h_out must be 16-byte aligned");
    const dim3 grid((hidden + 15) / 16, (batch + 15) / 16);
    for (int t = 0; t < frames; ++t)
        hipLaunchKernelGGL(lstm_step_kernel, grid, dim3(64 * LSTM_WAVES), 0, as_stream(stream), gates_ws, w_hh, cell_ws, h_out, batch, frames, hidden, t);
    return launch_status("nbasr_lstm_recurrence");
}

static inline int lstm_slices(int hidden) { return (hidden + LSTMP_HU - 1) / LSTMP_HU; }
static inline int lstm_kchunks_p(int hidden) { return ((hidden + 15) / 16 + LSTM_CHUNKS - 1) / LSTM_CHUNKS * LSTM_CHUNKS; }

extern "C" size_t nbasr_lstm_packed_whh_bytes(int hidden)
{
    if (hidden <= 0) return 0;
    return static_cast<size_t>(lstm_slices(hidden)) * lstm_kchunks_p(hidden) * 2 * 64 * sizeof(float4);
}

extern "C" int nbasr_lstm_pack_whh(const float* w_hh, void* packed, int hidden, nbasr_stream_t stream)
{
    clear_error();
    NBASR_REQUIRE(hidden > 0 && hidden % 4 == 0, NBASR_EALIGN, "nbasr_lstm_pack_whh: hidden=%d must be a positive multiple of 4", hidden);
    NBASR_REQUIRE(w_hh && packed, NBASR_ENULL, "nbasr_lstm_pack_whh: NULL pointer");
    NBASR_REQUIRE(aligned16(w_hh) && aligned16(packed), NBASR_EALIGN, "nbasr_lstm_pack_whh: buffers must be 16-byte aligned");
    hipLaunchKernelGGL(lstm_pack_whh_kernel, dim3(512), dim3(256), 0, as_stream(stream), w_hh, static_cast<float4*>(packed), hidden,
                       lstm_slices(hidden), lstm_kchunks_p(hidden));
    return launch_status("nbasr_lstm_pack_whh");
}

extern "C" int nbasr_lstm_recurrence_packed(const float* gates_ws, const void* packed_whh, float* cell_ws, float* h_out, int batch,
                                            int frames, int hidden, nbasr_stream_t stream)
{
    clear_error();
    const int rc = lstm_check("nbasr_lstm_recurrence_packed", batch, 4, frames, frames, hidden);
    if (rc != NBASR_OK) return rc;
    if (batch == 0 || frames == 0) return NBASR_OK;
    NBASR_REQUIRE(gates_ws && packed_whh && cell_ws && h_out, NBASR_ENULL, "nbasr_lstm_recurrence_packed: NULL pointer");
    NBASR_REQUIRE(aligned16(packed_whh) && aligned16(h_out), NBASR_EALIGN, "nbasr_lstm_recurrence_packed: packed_whh, h_out must be 16-byte aligned");
    const dim3 grid(lstm_slices(hidden), (batch + 15) / 16);
    for (int t = 0; t < frames; ++t)
        hipLaunchKernelGGL(lstm_step_packed_kernel, grid, dim3(64 * LSTM_WAVES), 0, as_stream(stream), gates_ws,
                           static_cast<const float4*>(packed_whh), cell_ws, h_out, batch, frames, hidden, lstm_kchunks_p(hidden), t);
    return launch_status("nbasr_lstm_recurrence_packed");
}

extern "C" int nbasr_lstm_forward_ln(const float* x, const float* w_ih, const float* w_hh, const float* b_ih,
                                     const float* b_hh, float* gates_ws, float* cell_ws, float* h_out, int batch,
                                     int c_in, int frames, int ld, int hidden, const nbasr_deferred_ln* ln,
                                     nbasr_stream_t stream)
{
    const int rc = nbasr_lstm_input_projection(x, w_ih, b_ih, b_hh, gates_ws, batch, c_in, frames, ld, hidden, ln, stream);
    if (rc != NBASR_OK) return rc;
    return nbasr_lstm_recurrence(gates_ws, w_hh, cell_ws, h_out, batch, frames, hidden, stream);
}

extern "C" int nbasr_lstm_forward(const float* x, const float* w_ih, const float* w_hh, const float* b_ih,
                                  const float* b_hh, float* gates_ws, float* cell_ws, float* h_out, int batch,
                                  int c_in, int frames, int ld, int hidden, nbasr_stream_t stream)
{
    return nbasr_lstm_forward_ln(x, w_ih, w_hh, b_ih, b_hh, gates_ws, cell_ws, h_out, batch, c_in, frames, ld, hidden, nullptr, stream);
}

extern "C" int nbasr_linear_head(const float* h, const float* w, const float* bias, float* logits, int rows,
                                 int features, int classes, nbasr_stream_t stream)
{
    clear_error();
    NBASR_REQUIRE(rows >= 0 && features > 0 && classes > 0 && classes <= 64, NBASR_EINVAL,
                  "nbasr_linear_head: rows=%d features=%d classes=%d (classes must be <= 64)", rows, features, classes);
    if (rows == 0) return NBASR_OK;
    NBASR_REQUIRE(h && w && bias && logits, NBASR_ENULL, "nbasr_linear_head: NULL pointer");
    NBASR_REQUIRE(features % 4 == 0 && aligned16(h) && aligned16(w), NBASR_EALIGN,
                  "nbasr_linear_head: features must be a multiple of 4 and h, w 16-byte aligned");
    if (rows == 0) return NBASR_OK;
    hipLaunchKernelGGL(head_kernel<false>, dim3((rows + 63) / 64), dim3(256), 0, as_stream(stream),
                       h, w, bias, logits, rows, features, classes, 0, 0, LnRef{nullptr, nullptr, nullptr});
    return launch_status("nbasr_linear_head");
}

extern "C" int nbasr_linear_head_bct_ln(const float* x, const float* w, const float* bias, float* logits, int batch,
                                        int features, int frames, int ld, int classes, const nbasr_deferred_ln* ln,
                                        nbasr_stream_t stream)
{
    clear_error();
    NBASR_REQUIRE(batch >= 0 && features > 0 && frames >= 0 && ld >= frames && classes > 0 && classes <= 64, NBASR_EINVAL,
                  "nbasr_linear_head_bct: bad sizes (classes must be <= 64)");
    if (batch == 0 || frames == 0) return NBASR_OK;
    NBASR_REQUIRE(x && w && bias && logits, NBASR_ENULL, "nbasr_linear_head_bct: NULL pointer");
    NBASR_REQUIRE(features % 4 == 0 && aligned16(w), NBASR_EALIGN, "nbasr_linear_head_bct: features must be a multiple of 4, w 16-byte aligned");
    NBASR_REQUIRE(!ln || (ln->stats && ln->gamma && ln->beta), NBASR_ENULL, "nbasr_linear_head_bct_ln: deferred LayerNorm needs stats, gamma and beta");
    const long long rows = static_cast<long long>(batch) * frames;
    if (rows == 0) return NBASR_OK;
    hipLaunchKernelGGL(head_kernel<true>, dim3(static_cast<unsigned>((rows + 63) / 64)), dim3(256), 0, as_stream(stream),
                       x, w, bias, logits, static_cast<int>(rows), features, classes, frames, ld, ln_ref(ln, true));
    return launch_status("nbasr_linear_head_bct");
}

extern "C" int nbasr_linear_head_bct(const float* x, const float* w, const float* bias, float* logits, int batch,
                                     int features, int frames, int ld, int classes, nbasr_stream_t stream)
{
    return nbasr_linear_head_bct_ln(x, w, bias, logits, batch, features, frames, ld, classes, nullptr, stream);
}
