// nn.LSTM(1200 -> 500, batch_first, one layer, zero initial state) and the CTC head nn.Linear
// (reference model.py:100-103, forward model.py:118-124), gate order i, f, g, o, bias b_ih + b_hh.
//
//  1. input projection for ALL frames at once: one fp32-MFMA GEMM (gemm_conv.hip, transposed store)
//     gates[t][b][4H] = x[b][:, t] . W_ih^T + b_ih + b_hh      -- the reference's permute(0,2,1) is
//     folded into the operand loader, nothing is transposed in memory;
//  2. the recurrence, on a fragment-ordered copy of W_hh: one launch per frame (lstm_step_packed_kernel), or all frames in one
//     launch with W_hh resident in registers (lstm_seq_kernel).  A workgroup owns 8 hidden units x 16 utterances and all four
//     gates; its 8 waves split K = hidden eight ways (v_mfma_f32_16x16x4_f32, 16-byte operand loads: the 4 components of a
//     lane's float4 feed 4 consecutive MFMAs, a k-permutation applied identically to W_hh and h, which leaves the sums
//     unchanged), partial tiles are reduced through LDS and the gate nonlinearities, cell update and h store are fused behind
//     the reduction.
//  3. head: logits = h . W^T + b on 16x16x4 MFMA tiles (classes padded to 64 in registers only).
//
// The recurrence is latency-bound (frames dependent steps).
#include "common.h"

#include <cstdlib>
#include <mutex>
#include <vector>

namespace nbasr {

constexpr int NBASR_MAX_DEVICES = 64;

typedef float floatx4 __attribute__((ext_vector_type(4)));

int lstm_input_projection(const float* x, const float* w_ih, const float* b_ih, const float* b_hh, float* gates,
                          int batch, int c_in, int frames, int ld, int rows4h, LnRef ln, hipStream_t stream);

__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }

// block 512 = 8 waves, each owning one eighth of K per round
constexpr int LSTM_WAVES = 8;
constexpr int LSTM_CHUNKS = 4;      // 16-deep k chunks per wave per round (8 waves x 4 x 16 = 512 >= hidden 500)

// ---- the recurrence on a fragment-ordered copy of w_hh -------------------------------------------------------------------
// Stamps on the first version of the step kernel (16 hidden units per workgroup, operands read from the (4H, H) matrix as it lies):
// its 160 operand wave-loads per workgroup (each touching 16 rows x 64 B) needed 4 100 - 8 000 cycles to land -- the L2 -> CU fill
// rate (~20 B/clk per CU), not latency -- while half of the CUs had no workgroup at all.  So (1) the weight is re-laid-out ONCE so
// that every wave-load is 1 KiB contiguous, and (2) a workgroup owns 8 hidden units (252 workgroups at H = 500, B = 64: 96 KiB of
// operands each instead of 160).  MFMA tile rows are (hidden unit, gate) = (row >> 2, row & 3), so the four gates of a (unit,
// utterance) pair sit in the four accumulator registers of ONE lane (C/D row = (lane >> 4) * 4 + reg): after the cross-wave K
// reduction the cell update needs no further exchange.
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
    float* __restrict__ cell, float* h_out, int batch, int frames, int hidden, int kchunks_p, int t, int prio)
{
    __shared__ float red[LSTM_WAVES][2][4][64];    // [wave][row tile][reg = gate][lane]
    // The per-frame chain runs on a side stream beside the next batch's encoder (executor.py): its waves share compute units with
    // encoder waves, and every frame waits for the slowest of its workgroups.  At <= 32 utterances per GPU that chain is (close to)
    // the critical path, so its waves take issue priority over whatever else the unit holds (round 4, same box, alternating runs:
    // 4 467 -> 4 635 utterances/s at 8, 6 483 -> 7 115 at 16, 8 635 -> 8 861 at 32); at 64 the encoder is the critical path and the
    // priority costs 0.8 %, so the launcher leaves it off there.  The one-launch recurrence gains nothing from it (measured).
    if (prio) __builtin_amdgcn_s_setprio(3);

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

// ---- the whole recurrence in ONE launch ------------------------------------------------------------------------------------------
// One launch per frame pays, per frame, a kernel boundary (1.5-1.9 us) and the L2 -> CU fill of the workgroup's 64 KiB slice of w_hh
// (~2 us at ~20 B/clk per CU) for 0.85 us of matrix work.  Here the grid of lstm_step_packed_kernel stays resident for all frames:
//   * a workgroup = (slice of 8 hidden units, tile of 16 utterances), its w_hh fragments loaded ONCE into registers (32 per lane),
//     its cell state kept in registers;
//   * h_t is exchanged through a double-buffered image in fragment order, hx[t & 1][k quad][utterance] (float4 = 4 consecutive k),
//     so that a wave's operand load is 1 KiB contiguous.  Utterance tiles are independent: only the (<= 64) slices of ONE tile
//     depend on each other;
//   * hand-off WITHOUT flags: the payload carries its own tag.  |h| <= 1, so bit 30 of every exchanged fp32 is zero: the producer sets
//     it to the tag of the step (lstm_tag: it flips every time a slot is rewritten, and the zeroed workspace never matches the first
//     one), in EVERY dword of the 16-byte granule, and stores the granule write-through (`sc1`, one store instruction per 16 bytes).
//     A consumer wave polls ITS OWN four operand granules with 16-byte `sc1` buffer loads (never served by the CU's L1) until every dword
//     of every lane shows the wanted tag -- a torn or stale granule fails the test in at least one dword -- then clears the bit and
//     multiplies.  No drain of the producer's stores, no flag store, no poll of a second location, no barrier around the loads: one
//     fabric round trip per step when the data are there (the flag form this replaces needed three: 4.2 us per frame).  A value that
//     cannot be an LSTM output (|h| >= 2, Inf, NaN) travels as the marker 1.5 and is decoded as NaN, so a diverged input stays visible.
//     Two images suffice: a workgroup can only publish step t + 1 (overwriting image (t + 1) & 1) after it has consumed every slice's
//     step t, which each slice published after consuming everybody's step t - 1;
//   * one workgroup barrier per step (the K partials of the 8 waves, double-buffered by step parity);
//   * every spin is bounded (1 s of the constant 100 MHz clock): on a timeout the workgroup raises the status word, fills the rest of
//     its h rows with NaN and leaves; nbasr_lstm_seq_status reports it.
// Same MFMA sequence per wave, same order of the eight wave partials and the same gate arithmetic as lstm_step_packed_kernel:
// bit-identical h.  Needs hidden <= 512 (all of K in ONE round of the 8 waves' 4 chunks) and <= 256 workgroups (co-residency).
constexpr int LSTMS_QUADS = LSTM_WAVES * LSTM_CHUNKS * 4;    // k quads of an exchange image (128: hidden <= 512)
constexpr int LSTMS_IMAGE_FLOATS = LSTMS_QUADS * 16 * 4;     // one image of one tile (32 KiB)
constexpr int LSTMS_HEADER_WORDS = 64;                       // [0] status
constexpr unsigned long long LSTMS_TIMEOUT_TICKS = 100000000ull;

typedef unsigned lstm_u4 __attribute__((ext_vector_type(4)));

// bit 30 of every dword of an exchanged granule carries the tag of its step: |h| <= 1 leaves that bit of an fp32 zero.  A slot of the
// double-buffered image is rewritten every second step, and the tag flips each time (the zeroed workspace never matches the first one)
__device__ __forceinline__ unsigned lstm_tag(int t) { return (static_cast<unsigned>((t >> 1) + 1) & 1u) << 30; }

__global__ __launch_bounds__(64 * LSTM_WAVES) void lstm_seq_kernel(
    const float* __restrict__ gates_in,   // (frames, batch, 4*hidden)
    const float4* __restrict__ wp,        // packed w_hh
    float* __restrict__ cell, float* __restrict__ h_out, unsigned* status, float* hx,
    int batch, int frames, int hidden, int kchunks_p, int flags)
{
    __shared__ float4 red[2][LSTM_WAVES][2][64];  // [step parity][wave][row tile][lane] -> the lane's 4 accumulator registers (= gates)
    __shared__ int stop;
    // NBASR_LSTM_SEQ_INJECT_FAULT (tests): the first slice of every tile never starts -- exactly what a grid that is not co-resident
    // looks like to the other slices, which time out, raise the status word and leave
    if ((flags & NBASR_LSTM_SEQ_INJECT_FAULT) && blockIdx.x == 0) return;

    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i16 = lane & 15, kq = lane >> 4;
    const int slice = blockIdx.x, tile = blockIdx.y;
    const int b0 = tile * 16;
    if (threadIdx.x == 0) stop = 0;

    float4 wv[LSTM_CHUNKS][2];
    {   // (a narrow layer has fewer chunks than the 8 waves cover: the surplus waves hold zeros and multiply the image's zero quads)
        const bool mine = wave * LSTM_CHUNKS < kchunks_p;
        const float4* wslice = wp + (static_cast<size_t>(slice) * kchunks_p + (mine ? wave * LSTM_CHUNKS : 0)) * 2 * 64 + lane;
#pragma unroll
        for (int c = 0; c < LSTM_CHUNKS; ++c) {
            wv[c][0] = mine ? wslice[c * 128] : make_float4(0.f, 0.f, 0.f, 0.f);
            wv[c][1] = mine ? wslice[c * 128 + 64] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }

    // epilogue role of waves 0 and 1: wave m owns row tile m -- lane = (unit kq of the tile, utterance i16), its 4 gates are the 4
    // accumulator registers.  Every wave issues the (clamped, branch-free) gate loads; only the epilogue waves use them.
    const int eu = slice * LSTMP_HU + (wave & 1) * 4 + kq, eb = b0 + i16;
    const bool e_ok = eu < hidden && eb < batch;
    const size_t gate_off = static_cast<size_t>(min(eb, batch - 1)) * (4 * hidden) + min(eu, hidden - 1);
    const int quad = slice * 2 + (wave & 1);
    const bool q_ok = kq == 0 && quad * 4 < hidden && eb < batch;      // lanes 0..15 store the tile's 4 units of one utterance
    float c_state = 0.f;

    float* const image = hx + static_cast<size_t>(tile) * 2 * LSTMS_IMAGE_FLOATS;
    const __amdgpu_buffer_rsrc_t hxr = __builtin_amdgcn_make_buffer_rsrc(image, 0, 2 * LSTMS_IMAGE_FLOATS * 4, 0x00020000);
    // the granules this lane consumes: utterance i16 of the tile, k quads (wave * 4 + c) * 4 + kq; those outside the batch / the layer
    // are never written and read as zeros
    bool live[LSTM_CHUNKS];
#pragma unroll
    for (int c = 0; c < LSTM_CHUNKS; ++c) live[c] = eb < batch && ((wave * LSTM_CHUNKS + c) * 4 + kq) * 4 < hidden;
    __syncthreads();

    int t = 0;
    for (; t < frames; ++t) {
        float pre[4];
        {
            const float* gin = gates_in + static_cast<size_t>(t) * batch * (4 * hidden) + gate_off;
#pragma unroll
            for (int g = 0; g < 4; ++g) pre[g] = gin[g * hidden];
        }
        floatx4 acc[2] = {floatx4{0.f, 0.f, 0.f, 0.f}, floatx4{0.f, 0.f, 0.f, 0.f}};
        if (t > 0) {
            // every wave polls ITS OWN operand granules of h_(t-1) until each dword carries that step's tag: no flag, no drain on the
            // producer's side, one fabric round trip per step when the data are already there
            const int img = ((t - 1) & 1) * LSTMS_QUADS;
            const unsigned want = lstm_tag(t - 1);
            floatx4 hr[LSTM_CHUNKS];
            const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
            bool ok = false;
            for (;;) {
                // The loads are inline assembly: a spin must really re-load every time (nothing for the optimiser to hoist or merge).
                // (Elements are taken with __float_as_uint: __builtin_bit_cast(unsigned, v[e]) of an ext-vector ELEMENT reads element 0
                // whatever e is -- hipcc, ROCm 7.2; tools/ubench/x3/bit_cast_of_vector_element.hip.)
#pragma unroll
                for (int c = 0; c < LSTM_CHUNKS; ++c) {
                    const int k4 = (wave * LSTM_CHUNKS + c) * 4 + kq;
                    const float* gp = image + ((img + k4) * 16 + i16) * 4;
                    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(hr[c]) : "v"(gp) : "memory");     // sc1: never served by this CU's L1
                }
                // all four loads in flight together, then ONE wait
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(hr[0]), "+v"(hr[1]), "+v"(hr[2]), "+v"(hr[3]) :: "memory");
                bool mine = true;
#pragma unroll
                for (int c = 0; c < LSTM_CHUNKS; ++c) {
                    unsigned tags = 0x40000000u;
#pragma unroll
                    for (int e = 0; e < 4; ++e) tags &= __float_as_uint(hr[c][e]) ^ ~want;       // bit 30 stays set while every dword's tag == want
                    mine = mine && (!live[c] || (tags & 0x40000000u) != 0);
                }
                ok = __all(mine);
                if (ok || __builtin_amdgcn_s_memrealtime() - t_start > LSTMS_TIMEOUT_TICKS) break;
                __builtin_amdgcn_s_sleep(1);
            }
            if (!ok && lane == 0) { stop = 1; __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
            float4 hv[LSTM_CHUNKS];
#pragma unroll
            for (int c = 0; c < LSTM_CHUNKS; ++c) {
                float d[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    unsigned u = __float_as_uint(hr[c][e]) & ~0x40000000u;
                    if ((u & 0x7fffffffu) == 0x3fc00000u) u = 0x7fc00000u;                  // the marker of a non-finite h
                    d[e] = live[c] ? __builtin_bit_cast(float, u) : 0.f;
                }
                hv[c] = make_float4(d[0], d[1], d[2], d[3]);
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
        const int par = t & 1;
#pragma unroll
        for (int m = 0; m < 2; ++m) red[par][wave][m][lane] = make_float4(acc[m][0], acc[m][1], acc[m][2], acc[m][3]);
        // ONE barrier per step.  The partials are double-buffered by step parity: a wave can only write those of step t + 2 after the
        // barrier of step t + 1, which the epilogue waves reach after they have read those of step t
        __syncthreads();
        if (stop) break;                          // (LDS, read behind the barrier; workgroup-uniform: a wave that timed out still came here)
        if (wave < 2) {
            float4 s = red[par][0][wave][lane];
#pragma unroll
            for (int w = 1; w < LSTM_WAVES; ++w) { const float4 r = red[par][w][wave][lane]; s.x += r.x; s.y += r.y; s.z += r.z; s.w += r.w; }
            pre[0] += s.x; pre[1] += s.y; pre[2] += s.z; pre[3] += s.w;
            const float c_new = sigmoidf_(pre[1]) * c_state + sigmoidf_(pre[0]) * tanhf(pre[2]);
            const float h_new = sigmoidf_(pre[3]) * tanhf(c_new);
            c_state = c_new;
            const float4 hq = make_float4(__shfl(h_new, i16), __shfl(h_new, i16 + 16), __shfl(h_new, i16 + 32), __shfl(h_new, i16 + 48));
            if (q_ok) {
                *reinterpret_cast<float4*>(h_out + (static_cast<size_t>(eb) * frames + t) * hidden + quad * 4) = hq;
                const unsigned tag = lstm_tag(t);
                lstm_u4 bits = {__builtin_bit_cast(unsigned, hq.x), __builtin_bit_cast(unsigned, hq.y), __builtin_bit_cast(unsigned, hq.z),
                                __builtin_bit_cast(unsigned, hq.w)};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (bits[e] & 0x40000000u) bits[e] = (bits[e] & 0x80000000u) | 0x3fc00000u;   // |h| >= 2, Inf, NaN cannot be an LSTM output: marker 1.5
                    bits[e] |= tag;
                }
                __builtin_amdgcn_raw_buffer_store_b128(bits, hxr, (((t & 1) * LSTMS_QUADS + quad) * 16 + i16) * 16, 0, 16);   // write-through, 16 bytes at once
            }
        }
    }
    if (t < frames) {                             // timed out: make the failure visible in the output too
        if (wave < 2 && q_ok) {
            const float nan = __builtin_nanf("");
            for (int u = t; u < frames; ++u)
                *reinterpret_cast<float4*>(h_out + (static_cast<size_t>(eb) * frames + u) * hidden + quad * 4) = make_float4(nan, nan, nan, nan);
        }
        return;
    }
    if (wave < 2 && e_ok) cell[static_cast<size_t>(eb) * hidden + eu] = c_state;
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
    const int prio = batch <= 32;          // issue priority for the chain's waves where the chain is the critical path (see the kernel)
    struct Ctx { dim3 grid; hipStream_t s; const float* gates; const float4* w; float* cell; float* h; int batch, frames, hidden, prio; };
    Ctx ctx{grid, as_stream(stream), gates_ws, static_cast<const float4*>(packed_whh), cell_ws, h_out, batch, frames, hidden, prio};
    // the chain of `frames` dependent launches, replayed as ONE cached graph per (buffers, shape, device) where the call recurs (common.h)
    const ChainKey key{{gates_ws, packed_whh, cell_ws, h_out, nullptr}, {batch, frames, hidden, 32, 0}};
    return replay_chain(ctx.s, key, "nbasr_lstm_recurrence_packed", [](void* p) {
        const Ctx& c = *static_cast<const Ctx*>(p);
        for (int t = 0; t < c.frames; ++t)
            hipLaunchKernelGGL(lstm_step_packed_kernel, c.grid, dim3(64 * LSTM_WAVES), 0, c.s, c.gates, c.w, c.cell, c.h, c.batch, c.frames, c.hidden,
                               lstm_kchunks_p(c.hidden), t, c.prio);
    }, &ctx);
}

// Workgroups of lstm_seq_kernel the CURRENT device holds at once (not a constant -- a partitioned (CPX) or CU-masked
// MI355X, or a smaller part, holds fewer than 256): compute units x the kernel's occupancy per unit, asked once per device.
// Without a device (host-only callers: tests of the size functions) the full part's 256.
static long lstm_seq_resident_workgroups()
{
    static std::mutex m;
    static long cached[NBASR_MAX_DEVICES] = {};
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess || device < 0 || device >= NBASR_MAX_DEVICES) { (void)hipGetLastError(); return 256; }
    std::lock_guard<std::mutex> lock(m);
    if (cached[device] == 0) {
        int cus = 0, per_cu = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, lstm_seq_kernel, 64 * LSTM_WAVES, 0) != hipSuccess || cus <= 0 || per_cu <= 0) {
            (void)hipGetLastError();
            return 256;
        }
        cached[device] = static_cast<long>(cus) * per_cu;
    }
    return cached[device];
}

static bool lstm_seq_fits(int batch, int hidden)
{
    if (batch <= 0 || hidden <= 0 || hidden % 4) return false;
    if (lstm_kchunks_p(hidden) > LSTM_WAVES * LSTM_CHUNKS) return false;
    // every workgroup resident at once; never more than one per CU of a full part (two such grids are chained, see below)
    const long grid = static_cast<long>(lstm_slices(hidden)) * ((batch + 15) / 16);
    return grid <= 256 && grid <= lstm_seq_resident_workgroups();
}

extern "C" size_t nbasr_lstm_seq_workspace_bytes(int batch, int hidden)
{
    if (!lstm_seq_fits(batch, hidden)) return 0;
    const size_t tiles = (batch + 15) / 16;
    return LSTMS_HEADER_WORDS * sizeof(unsigned) + tiles * 2 * LSTMS_IMAGE_FLOATS * sizeof(float);
}

extern "C" int nbasr_lstm_recurrence_seq(const float* gates_ws, const void* packed_whh, float* cell_ws, float* h_out, void* seq_ws,
                                         int batch, int frames, int hidden, int flags, nbasr_stream_t stream)
{
    clear_error();
    const int rc = lstm_check("nbasr_lstm_recurrence_seq", batch, 4, frames, frames, hidden);
    if (rc != NBASR_OK) return rc;
    if (batch == 0 || frames == 0) return NBASR_OK;
    NBASR_REQUIRE(gates_ws && packed_whh && cell_ws && h_out && seq_ws, NBASR_ENULL, "nbasr_lstm_recurrence_seq: NULL pointer");
    NBASR_REQUIRE(aligned16(packed_whh) && aligned16(h_out) && aligned16(seq_ws), NBASR_EALIGN,
                  "nbasr_lstm_recurrence_seq: packed_whh, h_out, seq_ws must be 16-byte aligned");
    NBASR_REQUIRE(lstm_seq_fits(batch, hidden), NBASR_EINVAL,
                  "nbasr_lstm_recurrence_seq: batch=%d hidden=%d does not fit one resident grid (hidden <= 512, ceil(hidden/8) * ceil(batch/16) "
                  "<= 256 workgroups); use nbasr_lstm_recurrence_packed", batch, hidden);
    const size_t tiles = (batch + 15) / 16;
    unsigned* const words = static_cast<unsigned*>(seq_ws);
    float* const hx = reinterpret_cast<float*>(words + LSTMS_HEADER_WORDS);
    // Two of these grids fit the chip together, three do not, and a grid whose workgroups are only partly resident waits for peers that
    // cannot start: launches of this kernel from different streams of the process are therefore chained, stream-ordered (each waits for
    // the event behind the previous one; no host synchronisation).  A stream under capture cannot take part in that chain.
    static std::mutex chain_mutex;
    static hipEvent_t chain_done[NBASR_MAX_DEVICES] = {};
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    hipError_t e = hipStreamIsCapturing(as_stream(stream), &capturing);
    NBASR_REQUIRE(e == hipSuccess && capturing == hipStreamCaptureStatusNone, NBASR_EINVAL,
                  "nbasr_lstm_recurrence_seq: the stream is being captured; a captured graph uses nbasr_lstm_recurrence_packed");
    int device = 0;
    e = hipGetDevice(&device);
    NBASR_REQUIRE(e == hipSuccess && device >= 0 && device < NBASR_MAX_DEVICES, NBASR_EINVAL, "nbasr_lstm_recurrence_seq: device %d out of range", device);
    std::lock_guard<std::mutex> lock(chain_mutex);
    hipEvent_t& done = chain_done[device];
    if (done == nullptr) e = hipEventCreateWithFlags(&done, hipEventDisableTiming);
    else e = hipStreamWaitEvent(as_stream(stream), done, 0);
    if (e != hipSuccess) { set_error("nbasr_lstm_recurrence_seq: %s", hipGetErrorString(e)); return static_cast<int>(e); }
    zero_async(seq_ws, nbasr_lstm_seq_workspace_bytes(batch, hidden), as_stream(stream));
    // A COOPERATIVE launch where the device offers it: the runtime then refuses a grid it cannot hold at once (instead of starting part
    // of it) and dispatches it as a unit.  What neither form can promise is that ANOTHER process leaves the compute units alone -- that
    // is what the bounded waits and the status word are for, and the executor reads the word behind every such launch (executor.py).
    static int coop[NBASR_MAX_DEVICES] = {};         // 0 unknown, 1 cooperative, -1 plain
    if (coop[device] == 0) {
        int ok = 0;
        coop[device] = (hipDeviceGetAttribute(&ok, hipDeviceAttributeCooperativeLaunch, device) == hipSuccess && ok) ? 1 : -1;
    }
    const dim3 grid(lstm_slices(hidden), static_cast<unsigned>(tiles)), block(64 * LSTM_WAVES);
    if (coop[device] > 0) {
        const float4* pw = static_cast<const float4*>(packed_whh);
        int kchunks = lstm_kchunks_p(hidden);
        unsigned* words_arg = words; float* hx_arg = hx;
        void* kargs[] = {&gates_ws, &pw, &cell_ws, &h_out, &words_arg, &hx_arg, &batch, &frames, &hidden, &kchunks, &flags};
        e = hipLaunchCooperativeKernel(reinterpret_cast<const void*>(lstm_seq_kernel), grid, block, kargs, 0, as_stream(stream));
        if (e != hipSuccess) {
            (void)hipGetLastError();
            set_error("nbasr_lstm_recurrence_seq: cooperative launch of %u x %u workgroups refused: %s (use nbasr_lstm_recurrence_packed)",
                      grid.x, grid.y, hipGetErrorString(e));
            return static_cast<int>(e);
        }
    } else {
        hipLaunchKernelGGL(lstm_seq_kernel, grid, block, 0, as_stream(stream), gates_ws, static_cast<const float4*>(packed_whh), cell_ws, h_out,
                           words, hx, batch, frames, hidden, lstm_kchunks_p(hidden), flags);
    }
    e = hipEventRecord(done, as_stream(stream));
    if (e != hipSuccess) { set_error("nbasr_lstm_recurrence_seq: hipEventRecord: %s", hipGetErrorString(e)); return static_cast<int>(e); }
    return launch_status("nbasr_lstm_recurrence_seq");
}

extern "C" int nbasr_lstm_seq_status(const void* seq_ws, nbasr_stream_t stream)
{
    clear_error();
    NBASR_REQUIRE(seq_ws, NBASR_ENULL, "nbasr_lstm_seq_status: NULL pointer");
    unsigned word = 0;
    hipError_t e = hipMemcpyAsync(&word, seq_ws, sizeof(word), hipMemcpyDeviceToHost, as_stream(stream));
    if (e == hipSuccess) e = hipStreamSynchronize(as_stream(stream));
    if (e != hipSuccess) { set_error("nbasr_lstm_seq_status: %s", hipGetErrorString(e)); return static_cast<int>(e); }
    NBASR_REQUIRE(word == 0, NBASR_EINVAL, "nbasr_lstm_seq_status: a step of the one-launch recurrence timed out waiting for its peers "
                  "(status %u): the grid was not co-resident", word);
    return NBASR_OK;
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

extern "C" int nbasr_linear_head_bct(const float* x, const float* w, const float* bias, float* logits, int batch,
                                     int features, int frames, int ld, int classes, const nbasr_deferred_ln* ln,
                                     nbasr_stream_t stream)
{
    clear_error();
    NBASR_REQUIRE(batch >= 0 && features > 0 && frames >= 0 && ld >= frames && classes > 0 && classes <= 64, NBASR_EINVAL,
                  "nbasr_linear_head_bct: bad sizes (classes must be <= 64)");
    if (batch == 0 || frames == 0) return NBASR_OK;
    NBASR_REQUIRE(x && w && bias && logits, NBASR_ENULL, "nbasr_linear_head_bct: NULL pointer");
    NBASR_REQUIRE(features % 4 == 0 && aligned16(w), NBASR_EALIGN, "nbasr_linear_head_bct: features must be a multiple of 4, w 16-byte aligned");
    NBASR_REQUIRE(!ln || (ln->stats && ln->gamma && ln->beta), NBASR_ENULL, "nbasr_linear_head_bct: deferred LayerNorm needs stats, gamma and beta");
    const long long rows = static_cast<long long>(batch) * frames;
    if (rows == 0) return NBASR_OK;
    hipLaunchKernelGGL(head_kernel<true>, dim3(static_cast<unsigned>((rows + 63) / 64)), dim3(256), 0, as_stream(stream),
                       x, w, bias, logits, static_cast<int>(rows), features, classes, frames, ld, ln_ref(ln, true));
    return launch_status("nbasr_linear_head_bct");
}

