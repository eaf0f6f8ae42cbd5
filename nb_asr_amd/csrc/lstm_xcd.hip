// The LSTM recurrence (reference model.py:100, 118-121: nn.LSTM(1200 -> 500), zero initial state, gates i, f, g, o) as ONE resident
// launch whose per-frame exchange never leaves an XCD, on the 16-bit matrix cores (round 6).
//
// Why.  The recurrence is `frames` dependent steps of a tiny GEMM (gates(2000) x utterances(16) x hidden(500)).  Rounds 2-5 ran it
// as one launch per frame (4.4-6.5 us per frame) or as a resident grid of 252 workgroups over the whole chip that exchange h_t in
// tagged write-through granules (3.1-3.8 us per frame: ~0.85 of fp32 MFMA, the rest the fabric round trip of an `sc1` store that
// drops its line from L2 + an `sc1` load that has to fetch it from the memory side, and an 8-way reduction through LDS).  Here:
//   * a tile of 16 utterances is owned by the (<= 32) workgroups of ONE XCD: workgroup = 16 hidden units x 4 gates = 4 MFMA row
//     tiles, one per wave, each wave holding ITS rows of w_hh for all of K = hidden <= 512 in registers (128 per lane).  A wave's
//     accumulators are complete sums: no reduction across waves, the gate arithmetic follows in the same registers;
//   * producers and consumers of a tile share one L2, so h_t is published with PLAIN stores (the line stays in that L2) behind
//     which each storing wave drains (`s_waitcnt vmcnt(0)`: L2 has the bytes) and sets its flag word; a consumer wave polls the
//     tile's 128 flag words (512 bytes) with `sc1` loads -- they bypass the CU's L1 and are served by L2 -- and then brings its
//     quarter of the 32 KiB image into LDS by LDS-DMA (`global_load_lds_dwordx4 sc1`: no registers, no LDS-store issue), one
//     workgroup barrier, and every wave reads all of h as ready-made B fragments.  One L2 round trip for the flag, one for the data,
//     per frame; nothing crosses the fabric (/opt/skills/guides/MI355X_MICROARCH.md, "stores of each flavour", hand-off rows);
//   * placement is not assumed, it is READ: a workgroup takes its XCD from HW_REG_XCC_ID and its slice from a per-XCD arrival ticket;
//     the first arrival of an XCD claims whole tiles from a global ticket once its XCD has all its slices, publishes the claim, and
//     its peers follow.  Whatever the dispatcher does, the workgroups that exchange a tile's h ARE on one XCD; an XCD that never
//     collects its slices claims nothing; surplus workgroups leave at once.  HIP's round-robin dealing of workgroups over the XCDs
//     (observed, not promised) only decides how many XCDs take part.  The last workgroup to leave checks that every tile was
//     computed and raises the status word otherwise;
//   * the recurrent product runs on v_mfma_f32_16x16x32_f16 with fp32-accurate operands: w_hh (scaled by one power of two) and h
//     (|h| <= 1) are two fp16 terms each, value = hi + lo' * 2^-11, and a product is hi*hi (first accumulator) + (hi*lo' + lo'*hi)
//     (second accumulator, folded in with its 2^-11 at the end): 3 MFMAs per 32 k, 1/5 of the fp32 MFMA time.  The dropped lo*lo
//     term is <= 2^-24 of the product; the representation error of an operand is 2^-23 relative (as in the dense convs and the
//     input projection, gemm_conv_split.hip) -- an fp32 evaluation in another summation order, not a narrower one (measured
//     against float64: tests/test_lstm_xcd_gpu.py);
//   * h_t is stored as it is consumed: the image IS the B operand of the MFMA, [k-step][hi | lo'][lane] x 8 halves; a wave
//     writes the 4 units of its row tile as one 8-byte piece per utterance and term.  A non-finite h travels as lo' = NaN, which
//     every product it meets turns into NaN: a diverged utterance stays visible;
//   * the gates use the hardware exp2 / rcp (1 ulp each) -- sigmoid(x) = rcp(1 + exp2(-x log2 e)); tanh(x) = (1 - e) rcp(1 + e),
//     e = exp2(-2 |x| log2 e), and an odd polynomial below |x| = 0.35 where that form would cancel: ~55 vector instructions per
//     frame instead of the ~200 of the library calls, at <= 2 ulp;
//   * every wait is bounded (1 s of the 100 MHz clock); a timeout raises the status word (nbasr_lstm_seq_status, or the
//     executor's asynchronous read-back) and fills the rest of that slice's h rows with NaN.
#include "common.h"

#include <algorithm>
#include <cstdlib>
#include <mutex>

namespace nbasr {

typedef float xfloat4 __attribute__((ext_vector_type(4)));
typedef _Float16 xhalf8 __attribute__((ext_vector_type(8)));
typedef unsigned xuint4 __attribute__((ext_vector_type(4)));
typedef unsigned xuint2 __attribute__((ext_vector_type(2)));

constexpr int LX_WAVES = 4;                      // a wave = one MFMA row tile (4 hidden units x 4 gates) x all of K
constexpr int LX_UNITS = 16;                     // hidden units per workgroup
constexpr int LX_KSTEPS = 16;                    // hidden <= 512
constexpr int LX_MAX_SLICES = LX_KSTEPS * 32 / LX_UNITS;   // 32
constexpr int LX_IMAGE_BYTES = LX_KSTEPS * 2 * 64 * 16;    // one exchange image of one tile: [k-step][hi | lo'][lane] x 16 B = 32 KiB
constexpr int LX_FLAGS = LX_MAX_SLICES * LX_WAVES;         // flag words of one tile (one per publishing wave): 128
constexpr int LX_TILE_BYTES = 2 * LX_IMAGE_BYTES + LX_FLAGS * 4;      // two images (step parity) + the flags
constexpr int LX_MAX_XCD = 16;
constexpr int LX_MAX_TILES = 256;                // batch <= 4096
// header of the workspace (32-bit words); [0] is the status word nbasr_lstm_seq_status reads
constexpr int LX_W_STATUS = 0, LX_W_NEXT_TILE = 1, LX_W_TILES_DONE = 2, LX_W_EXITS = 3;
constexpr int LX_W_ARRIVALS = 16;                // + xcd * 16 (a counter per 64-byte line)
constexpr int LX_W_CLAIMS = LX_W_ARRIVALS + LX_MAX_XCD * 16;          // + xcd * (LX_MAX_TILES + 16) + seq
constexpr int LX_CLAIM_STRIDE = LX_MAX_TILES + 16;
constexpr int LX_W_RANGE = LX_W_CLAIMS + LX_MAX_XCD * LX_CLAIM_STRIDE;       // + utterance: max finite |gate pre-activation| over all its frames (float bits)
constexpr int LX_HEADER_WORDS = LX_W_RANGE + LX_MAX_TILES * 16;
constexpr unsigned LX_DONE = 0xffffffffu;
constexpr unsigned long long LX_TIMEOUT_TICKS = 100000000ull;         // 1 s of the constant 100 MHz clock
constexpr int LX_PACK_HEADER_BYTES = 256;        // packed w_hh: [0] 2^-e (float), [2] max |w| bits (pack-time scratch)

// Diagnostic build (-DNBASR_LX_STAMPS=1, tools/ubench/lstm_xcd_stamps.py): wave 0 of every slice of tile 0 stamps the shader clock at its
// phase boundaries of every frame into a region behind the tiles' images.  Never in the shipped library.
#ifndef NBASR_LX_STAMPS
#define NBASR_LX_STAMPS 0
#endif
constexpr int LX_STAMP_POINTS = 8, LX_STAMP_FRAMES = 256;
constexpr size_t LX_STAMP_BYTES = NBASR_LX_STAMPS ? static_cast<size_t>(32) * LX_STAMP_FRAMES * LX_STAMP_POINTS * 8 : 0;
#if NBASR_LX_STAMPS
#define LX_STAMP(i) do { if (stamping) st[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define LX_STAMP(i) do { } while (0)
#endif

// sigmoid and tanh on the hardware exp2 / rcp (v_exp_f32, v_rcp_f32: 1 ulp each).  Saturation is exact: exp2 -> 0 or +inf, rcp(inf) = 0.
__device__ __forceinline__ float lx_sigmoid(float v)
{
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.4426950408889634f));
}
__device__ __forceinline__ float lx_tanh(float v)
{
    const float a = fabsf(v);
    const float e = __builtin_amdgcn_exp2f(a * -2.8853900817779268f);           // exp(-2 |v|) in (0, 1]
    const float big = (1.0f - e) * __builtin_amdgcn_rcpf(1.0f + e);               // absolute error ~3e-8: fine from 0.35 up
    const float x2 = v * v;                                                       // below: the odd series through x^11 (next term 1.2e-8 relative at 0.35)
    float p = __builtin_fmaf(x2, -0.0088632355299021966f, 0.021869488536155203f);
    p = __builtin_fmaf(x2, p, -0.053968253968253971f);
    p = __builtin_fmaf(x2, p, 0.13333333333333333f);
    p = __builtin_fmaf(x2, p, -0.33333333333333331f);
    const float small = __builtin_fmaf(v * x2, p, v);
    return a < 0.35f ? small : __builtin_copysignf(big, v);                       // (NaN: a < 0.35 is false, big is NaN)
}

__device__ __forceinline__ unsigned lx_load_sc1(const unsigned* p)
{
    unsigned v;
    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void lx_store_sc1(unsigned* p, unsigned v)
{
    asm volatile("global_store_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" :: "v"(p), "v"(v) : "memory");
}

// ---- packing -------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lx_absmax_kernel(const float* __restrict__ w, size_t n, unsigned* header)
{
    float m = 0.f;
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += static_cast<size_t>(gridDim.x) * blockDim.x) m = fmaxf(m, finite_abs(w[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) atomicMax(header + 2, __float_as_uint(m));          // non-negative floats order like their bits
}

// packed[256 B header][slice][wave (= row tile)][k-step][part (hi, lo')][lane] x 16 B: the A fragment of v_mfma_f32_16x16x32_f16 for the
// row tile (row i = lane & 15: unit slice*16 + wave*4 + (i >> 2), gate i & 3) and k-step (k = kstep*32 + (lane >> 4)*8 + j)
__global__ __launch_bounds__(256) void lx_pack_kernel(const float* __restrict__ w_hh, unsigned char* __restrict__ packed, int hidden, int slices)
{
    float* const hdr = reinterpret_cast<float*>(packed);
    const float amax = __uint_as_float(reinterpret_cast<const unsigned*>(packed)[2]);
    int e = 0;
    if (amax > 0.f) { int ex; (void)frexpf(amax, &ex); e = 14 - ex; }                // amax * 2^e in [2^13, 2^14)
    const float scale = ldexpf(1.0f, e);
    if (blockIdx.x == 0 && threadIdx.x == 0) hdr[0] = ldexpf(1.0f, -e);
    xuint4* const out = reinterpret_cast<xuint4*>(packed + LX_PACK_HEADER_BYTES);
    const size_t total = static_cast<size_t>(slices) * LX_WAVES * LX_KSTEPS * 2 * 64;
    for (size_t idx = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; idx < total; idx += static_cast<size_t>(gridDim.x) * blockDim.x) {
        size_t r = idx;
        const int lane = r % 64; r /= 64;
        const int part = r % 2; r /= 2;
        const int ks = r % LX_KSTEPS; r /= LX_KSTEPS;
        const int wave = r % LX_WAVES; r /= LX_WAVES;
        const int slice = static_cast<int>(r);
        const int i = lane & 15, kq = lane >> 4;
        const int unit = slice * LX_UNITS + wave * 4 + (i >> 2), gate = i & 3;
        const int k0 = ks * 32 + kq * 8;
        unsigned short h[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = 0.f;
            if (unit < hidden && k0 + j < hidden) v = w_hh[(static_cast<size_t>(gate) * hidden + unit) * hidden + k0 + j] * scale;
            const _Float16 hi = static_cast<_Float16>(v);
            const _Float16 lo = static_cast<_Float16>((v - static_cast<float>(hi)) * 2048.0f);
            h[j] = __builtin_bit_cast(unsigned short, part == 0 ? hi : lo);
        }
        out[idx] = xuint4{static_cast<unsigned>(h[0]) | (static_cast<unsigned>(h[1]) << 16), static_cast<unsigned>(h[2]) | (static_cast<unsigned>(h[3]) << 16),
                          static_cast<unsigned>(h[4]) | (static_cast<unsigned>(h[5]) << 16), static_cast<unsigned>(h[6]) | (static_cast<unsigned>(h[7]) << 16)};
    }
}

// ---- the range of an utterance's h ------------------------------------------------------------------------------------------------
// Two fp16 terms hold 22 bits of a value -- down to an ABSOLUTE floor of 2^-35 (lo' = (h - hi) 2^11 as an fp16 subnormal).  A model whose
// activations have decayed (the reference's own initialisation drives the benchmark architecture to 1e-25) has h far below
// that, and such a model must still come out right RELATIVE to its own scale.  As the dense convolutions do with their input, h is therefore
// exchanged scaled by an exact power of two per UTTERANCE (an MFMA column: the factor leaves with one multiply of the sums; per utterance,
// so a result never depends on the batch it sits in): 2^k with k = clamp(-8 - floor(log2 gmax), 0, 96), gmax = the utterance's largest
// finite |input projection| over all frames and gates -- |h| <= |tanh c| and c sums bounded multiples of tanh(pre-activations), so h
// sits at or below a small multiple of that scale (up to ~frames x above it if a forget gate stays open: fp16 has 2^23 of room above).  An
// utterance of ordinary size (gmax >= 2^-8: every trained or He-initialised model) gets k = 0 and bits as without the scaling.
__global__ __launch_bounds__(256) void lx_gate_range_kernel(const float* __restrict__ gates, unsigned* __restrict__ range, int batch, int frames, int row)
{
    // a workgroup = one utterance x 8 frames; a thread's loads of the 8 frames are independent (all in flight together): the pass is one
    // read of the gate tensor at memory speed (first version, a loop over frames with two loads in flight per thread: 1.6 TB/s)
    const int b = blockIdx.y, t0 = blockIdx.x * 8, nq = row >> 2;
    float m = 0.f;
    for (int i = threadIdx.x; i < nq; i += blockDim.x) {
        float4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int t = min(t0 + j, frames - 1);
            v[j] = reinterpret_cast<const float4*>(gates + (static_cast<size_t>(t) * batch + b) * row)[i];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) m = fmaxf(fmaxf(m, fmaxf(finite_abs(v[j].x), finite_abs(v[j].y))), fmaxf(finite_abs(v[j].z), finite_abs(v[j].w)));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    __shared__ float s_m[4];
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
        if (m > 0.f) atomicMax(range + b, __float_as_uint(m));
    }
}
// (2^k, 2^-k) from the range word
__device__ __forceinline__ void lx_h_scale(unsigned range_bits, float& hs, float& hs_inv)
{
    const int k = min(max(119 - static_cast<int>((range_bits >> 23) & 0xffu), 0), 96);
    hs = __uint_as_float(static_cast<unsigned>(127 + k) << 23);
    hs_inv = __uint_as_float(static_cast<unsigned>(127 - k) << 23);
}

// ---- one row tile (4 hidden units x 4 gates) x all of K: the chain both recurrence kernels run, in this order ---------------------------
// acc0 += hi * hi ; acc1 += hi * lo' + lo' * hi, k-step after k-step.  (Measured on the way, 250 frames x 64 utterances: the fragment
// reads issued step by step, two registers deep -- what the scheduler makes of the plain loop -- 2 100 cycles per frame for this phase, a
// wave alone on its SIMD has nobody to hide an LDS round trip behind; all 32 reads first 1 630; this order 1 450.  The chain split into the
// wave's own quarter before the barrier and the rest behind it: slower, 580 + 1 420.)
__device__ __forceinline__ void lx_row_tile_product(const xuint4 (&h)[LX_KSTEPS][2][64], const xuint4 (&w)[LX_KSTEPS][2], int lane, xfloat4& acc0, xfloat4& acc1)
{
    xuint4 bfrag[LX_KSTEPS][2];
#pragma unroll
    for (int ks = 0; ks < LX_KSTEPS; ++ks) { bfrag[ks][0] = h[ks][0][lane]; bfrag[ks][1] = h[ks][1][lane]; }
#pragma unroll
    for (int ks = 0; ks < LX_KSTEPS; ++ks) {
        const xhalf8 hh = __builtin_bit_cast(xhalf8, bfrag[ks][0]), hl = __builtin_bit_cast(xhalf8, bfrag[ks][1]);
        const xhalf8 wh = __builtin_bit_cast(xhalf8, w[ks][0]), wl = __builtin_bit_cast(xhalf8, w[ks][1]);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, hh, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, hl, acc1, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, hh, acc1, 0, 0, 0);
    }
    // the order the scheduler is to keep: the reads of 6 k-steps ahead, then per k-step its 3 MFMAs and the reads of the step six ahead
    // (lgkmcnt counts to 15: twelve reads in flight are still told apart) -- the first MFMA starts one LDS round trip after the barrier,
    // the rest of the reads travel beside the matrix work
    __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);
#pragma unroll
    for (int ks = 0; ks < LX_KSTEPS; ++ks) {
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
        if (ks + 6 < LX_KSTEPS) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    }
}

// gates i, f, g, o of one (hidden unit, utterance): s = w_hh . h_(t-1) (the four accumulator registers), pre = the input projection
__device__ __forceinline__ float lx_cell_update(const xfloat4& s, const float (&pre)[4], float& c_state)
{
    const float p0 = s[0] + pre[0], p1 = s[1] + pre[1], p2 = s[2] + pre[2], p3 = s[3] + pre[3];
    const float c_new = lx_sigmoid(p1) * c_state + lx_sigmoid(p0) * lx_tanh(p2);
    c_state = c_new;
    return lx_sigmoid(p3) * lx_tanh(c_new);
}

// h as it is consumed: (fp16 hi << 16) | fp16 lo', lo' = (h - hi) * 2^11; a value that cannot be an LSTM output travels as lo' = NaN
__device__ __forceinline__ unsigned lx_split_h(float h_new, float hs)
{
    const float v = h_new * hs;                      // (the utterance's power of two: exact)
    const _Float16 hi = static_cast<_Float16>(v);
    const _Float16 lo = static_cast<_Float16>((v - static_cast<float>(hi)) * 2048.0f);
    const unsigned dw = (static_cast<unsigned>(__builtin_bit_cast(unsigned short, hi)) << 16) | __builtin_bit_cast(unsigned short, lo);
    return fabsf(h_new) <= 1.0f ? dw : 0x00007e00u;
}

// the 4 units of a row tile x one utterance as two 8-byte pieces of the image (4 hi halves, 4 lo' halves): lanes 0..15 store
__device__ __forceinline__ void lx_publish(unsigned dw, int n16, bool q_ok, unsigned char* dst)
{
    const unsigned d0 = static_cast<unsigned>(__shfl(static_cast<int>(dw), n16)), d1 = static_cast<unsigned>(__shfl(static_cast<int>(dw), n16 + 16));
    const unsigned d2 = static_cast<unsigned>(__shfl(static_cast<int>(dw), n16 + 32)), d3 = static_cast<unsigned>(__shfl(static_cast<int>(dw), n16 + 48));
    if (q_ok) {
        const xuint2 hq = {__builtin_amdgcn_perm(d1, d0, 0x07060302u), __builtin_amdgcn_perm(d3, d2, 0x07060302u)};
        const xuint2 lq = {__builtin_amdgcn_perm(d1, d0, 0x05040100u), __builtin_amdgcn_perm(d3, d2, 0x05040100u)};
        *reinterpret_cast<xuint2*>(dst) = hq;                         // PLAIN stores: the lines stay in this XCD's L2
        *reinterpret_cast<xuint2*>(dst + 1024) = lq;
    }
}

// where wave `wave` of slice `slice` writes its pieces for utterance n16: k-step slice/2, k-octet 2*(slice&1) + (wave>>1), halves 4*(wave&1) .. +3
__device__ __forceinline__ int lx_publish_offset(int slice, int wave, int n16)
{
    return (((slice >> 1) * 2 * 64) + (2 * (slice & 1) + (wave >> 1)) * 16 + n16) * 16 + (wave & 1) * 8;
}

// ---- one launch per frame, the SAME arithmetic -------------------------------------------------------------------------------------------
// What a pipelined tail runs beside the next batch's encoder (a resident grid there would fill whole XCDs for the length of the
// recurrence, and every encoder kernel has workgroups dealt to those XCDs), and what a plan demoted by a failed status word falls back
// to: frame t as its own launch, workgroup = (slice of 16 hidden units, tile of 16 utterances) as in lstm_xcd_kernel, the weights
// fetched from L2 per launch, h_(t-1) read from the image the previous launch wrote.  Same product chain, same gate arithmetic, same
// image: bit-identical h to the resident form.  Against the fp32 per-frame kernel (lstm.hip): a fifth of the matrix time and half the
// workgroups per frame (128 instead of 252 at 64 utterances).
// WPB = row tiles (waves) per workgroup, 1, 2 or 4: the decomposition does not touch the arithmetic (a row tile is one wave's chain
// whatever the workgroup) -- small batches take two-wave workgroups (63 per tile: the shortest frames), large ones four-wave workgroups
// (32 per tile: the image is fetched once per four row tiles, the least work beside the encoder).  A workgroup CAN walk several utterance
// tiles with one fetch of its weights (tiles_per_block); measured slower, see the launcher.
template <int WPB>
__global__ __launch_bounds__(64 * WPB) void lstm_step16_kernel(
    const float* __restrict__ gates_in, const unsigned char* __restrict__ wp, float* __restrict__ cell, float* __restrict__ h_out,
    unsigned char* tiles, int batch, int frames, int hidden, int t, int prio, int tiles_per_block)
{
    constexpr int MAX_TPB = 4;                             // utterance tiles a workgroup walks with ONE fetch of its weights
    __shared__ xuint4 htile[LX_KSTEPS][2][64];
    if (prio & 1) __builtin_amdgcn_s_setprio(3);
#ifdef NBASR_LX_TAIL_EXPERIMENT        // (tools/ubench/tail_cost.py --what: timing only, wrong results)
    if (prio & 4) return;                                  // the launches alone
#endif
    const int lane = threadIdx.x & 63, wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n16 = lane & 15, kq = lane >> 4;
    const int row_tile = blockIdx.x * WPB + wib;           // of the layer: 4 hidden units x 4 gates
    const int slice = row_tile >> 2, wave = row_tile & 3;  // as lstm_xcd_kernel names them (weights, image)
    const int n_tiles = (batch + 15) >> 4, tile0 = blockIdx.y * tiles_per_block;
    const int eu = row_tile * 4 + kq;
    const unsigned* const range = reinterpret_cast<const unsigned*>(tiles) - (LX_HEADER_WORDS - LX_W_RANGE);

    // everything that comes from far away is requested first, for all of the workgroup's tiles: the gate pre-activations and the cell
    // states (HBM / last-level cache), then the weights; the images follow tile by tile
    float pre[MAX_TPB][4], c_prev[MAX_TPB];
#pragma unroll
    for (int k = 0; k < MAX_TPB; ++k) {
        const int eb = (tile0 + k) * 16 + n16;
        const bool live = k < tiles_per_block && tile0 + k < n_tiles;
        const size_t gate_off = static_cast<size_t>(min(eb, batch - 1)) * (4 * hidden) + min(eu, hidden - 1);
#pragma unroll
        for (int g = 0; g < 4; ++g) pre[k][g] = live ? gates_in[static_cast<size_t>(t) * batch * (4 * hidden) + gate_off + g * hidden] : 0.f;
        c_prev[k] = (live && t > 0 && eu < hidden && eb < batch) ? cell[static_cast<size_t>(eb) * hidden + eu] : 0.f;
    }
    xuint4 wfrag[LX_KSTEPS][2];
#ifdef NBASR_LX_TAIL_EXPERIMENT
    if (prio & 2) {                                        // no weight fetch
#pragma unroll
        for (int ks = 0; ks < LX_KSTEPS; ++ks) { wfrag[ks][0] = xuint4{0, 0, 0, 0}; wfrag[ks][1] = xuint4{0, 0, 0, 0}; }
    } else
#endif
    if (t > 0) {
        // (a row tile beyond the layer -- the grid is rounded up to whole workgroups -- reads the zero rows the packer wrote up to the slice's end,
        // or, past the last slice, the last slice's: its sums are never stored)
        const int wslice = min(slice, (hidden + LX_UNITS - 1) / LX_UNITS - 1);
        const xuint4* src = reinterpret_cast<const xuint4*>(wp + LX_PACK_HEADER_BYTES) + (static_cast<size_t>(wslice) * LX_WAVES + wave) * (LX_KSTEPS * 2 * 64) + lane;
#pragma unroll
        for (int ks = 0; ks < LX_KSTEPS; ++ks) { wfrag[ks][0] = src[(ks * 2) * 64]; wfrag[ks][1] = src[(ks * 2 + 1) * 64]; }
    }
    const float winv = reinterpret_cast<const float*>(wp)[0];
#pragma unroll
    for (int k = 0; k < MAX_TPB; ++k) {
        const int tile = tile0 + k;
        if (k >= tiles_per_block || tile >= n_tiles) break;                           // (workgroup-uniform)
        unsigned char* const tile_ws = tiles + static_cast<size_t>(tile) * LX_TILE_BYTES;
        const int eb = tile * 16 + n16;
        const bool e_ok = eu < hidden && eb < batch;
        float hs, hs_inv;
        lx_h_scale(range[min(eb, batch - 1)], hs, hs_inv);
        xfloat4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        if (t > 0) {
            const unsigned char* img = tile_ws + ((t - 1) & 1) * LX_IMAGE_BYTES;
            if (k > 0) __syncthreads();                    // every wave has read the previous tile's image
            // (columns of utterances beyond the batch are not fetched: nothing reads their sums, and a part-filled tile -- 8 utterances, one
            // rank's share of the benchmark batch on 8 GPUs -- moves half the image)
            if (eb < batch) {
#pragma unroll
                for (int j = 0; j < 32 / WPB; ++j) {
                    const int chunk = wib * (32 / WPB) + j;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(img + chunk * 1024 + lane * 16),
                                                     (__attribute__((address_space(3))) void*)(reinterpret_cast<unsigned char*>(&htile[0][0][0]) + chunk * 1024), 16, 0, 0);
                }
            }
            __syncthreads();                               // (drains this wave's transfers, then the whole image is in LDS)
            lx_row_tile_product(htile, wfrag, lane, acc0, acc1);
        }
        const xfloat4 s = (acc0 + acc1 * 0.00048828125f) * (winv * hs_inv);
        float c_state = c_prev[k];
        const float h_new = lx_cell_update(s, pre[k], c_state);
        const bool q_ok = kq == 0 && row_tile * 4 < hidden && eb < batch;
        lx_publish(lx_split_h(h_new, hs), n16, q_ok, tile_ws + (t & 1) * LX_IMAGE_BYTES + lx_publish_offset(slice, wave, n16));
        if (e_ok) {
            h_out[(static_cast<size_t>(eb) * frames + t) * hidden + eu] = h_new;
            cell[static_cast<size_t>(eb) * hidden + eu] = c_state;
        }
    }
}

// ---- the recurrence in ONE launch ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64 * LX_WAVES) void lstm_xcd_kernel(
    const float* __restrict__ gates_in,     // (frames, batch, 4*hidden)
    const unsigned char* __restrict__ wp,   // packed w_hh (lx_pack_kernel)
    float* __restrict__ cell, float* __restrict__ h_out, unsigned* ws,
    int batch, int frames, int hidden, int slices, int n_tiles, int total_wgs, int flags)
{
    __shared__ xuint4 htile[2][LX_KSTEPS][2][64];          // [step parity][k-step][hi | lo'][lane]: h_(t-1) of the tile as B fragments (2 x 32 KiB)
    __shared__ unsigned s_role[4];                         // [0] xcd, [1] rank, [2] claim of the current round, [3] stop

    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n16 = lane & 15, kq = lane >> 4;
    unsigned char* const tiles = reinterpret_cast<unsigned char*>(ws + LX_HEADER_WORDS);

    // ---- who am I: XCD from the hardware register, slice from the XCD's arrival ticket ----
    if (threadIdx.x == 0) {
        const unsigned xcd = __builtin_amdgcn_s_getreg(20 | (31 << 11)) & 0xfu;        // HW_REG_XCC_ID
        unsigned rank = atomicAdd(ws + LX_W_ARRIVALS + xcd * 16, 1u);
        // NBASR_LSTM_SEQ_INJECT_FAULT (tests): the second arrival of every XCD takes its ticket and leaves -- what a compute unit taken
        // away by another process looks like to its peers: they wait for its flags, time out, raise the status word
        if ((flags & NBASR_LSTM_SEQ_INJECT_FAULT) && rank == 1) rank = 0x7fffffffu;
        s_role[0] = xcd; s_role[1] = rank; s_role[3] = 0;
    }
    __syncthreads();
    const unsigned xcd = s_role[0], rank = s_role[1];
    unsigned* const claims = ws + LX_W_CLAIMS + xcd * LX_CLAIM_STRIDE;
    const bool member = rank < static_cast<unsigned>(slices);
    const int slice = static_cast<int>(rank);

    xuint4 wfrag[LX_KSTEPS][2];                            // [k-step][hi, lo']: 128 registers, loaded with the first claimed tile
    bool have_w = false;
    float inv = 1.f;

    for (int seq = 0; member; ++seq) {
        // ---- which tile: the XCD's first arrival claims, the others follow its published claim ----
        if (threadIdx.x == 0) {
            unsigned claim = LX_DONE;
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            if (rank == 0) {
                bool complete = seq > 0;
                while (!complete) {                        // all slices of this XCD present?  (or nothing left to claim / out of time)
                    if (lx_load_sc1(ws + LX_W_ARRIVALS + xcd * 16) >= static_cast<unsigned>(slices)) { complete = true; break; }
                    if (lx_load_sc1(ws + LX_W_NEXT_TILE) >= static_cast<unsigned>(n_tiles)) break;
                    if (__builtin_amdgcn_s_memrealtime() - t0 > LX_TIMEOUT_TICKS) break;
                    __builtin_amdgcn_s_sleep(8);
                }
                if (complete) {
                    const unsigned tile = atomicAdd(ws + LX_W_NEXT_TILE, 1u);
                    if (tile < static_cast<unsigned>(n_tiles)) claim = tile + 1;
                }
                lx_store_sc1(claims + seq, claim);
            } else {
                for (;;) {
                    claim = lx_load_sc1(claims + seq);
                    if (claim != 0) break;
                    if (__builtin_amdgcn_s_memrealtime() - t0 > 2 * LX_TIMEOUT_TICKS) { claim = LX_DONE; break; }    // (the first arrival gives up after 1 s)
                    __builtin_amdgcn_s_sleep(8);
                }
            }
            s_role[2] = claim;
        }
        __syncthreads();                                   // (also: every wave has left the previous tile's last step)
        const unsigned claim = s_role[2];
        if (claim == LX_DONE) break;
        const int tile = static_cast<int>(claim) - 1;
        const int b0 = tile * 16;

        if (!have_w) {
            have_w = true;
            inv = reinterpret_cast<const float*>(wp)[0];
            const xuint4* src = reinterpret_cast<const xuint4*>(wp + LX_PACK_HEADER_BYTES) + (static_cast<size_t>(slice) * LX_WAVES + wave) * (LX_KSTEPS * 2 * 64) + lane;
#pragma unroll
            for (int ks = 0; ks < LX_KSTEPS; ++ks) { wfrag[ks][0] = src[(ks * 2) * 64]; wfrag[ks][1] = src[(ks * 2 + 1) * 64]; }
            // RESIDENT, in the accumulator half of the register file (an MFMA takes its A operand from there as well): without this the
            // compiler, short of architectural VGPRs (256) for 128 weight + 128 fragment registers, re-loads the `const __restrict__`
            // weights from memory in EVERY frame (32 KiB per wave and frame through L2: seen in the ISA, 2 100 cycles per frame)
#pragma unroll
            for (int ks = 0; ks < LX_KSTEPS; ++ks) { asm volatile("" : "+a"(wfrag[ks][0])); asm volatile("" : "+a"(wfrag[ks][1])); }
        }

        unsigned char* const tile_ws = tiles + static_cast<size_t>(tile) * LX_TILE_BYTES;
        unsigned* const tflags = reinterpret_cast<unsigned*>(tile_ws + 2 * LX_IMAGE_BYTES);
        // the flags this lane watches: words lane and lane + 64 of the tile's 128 (word = slice * 4 + wave); only those of slices that exist
        const bool watch0 = (lane >> 2) < slices, watch1 = ((lane + 64) >> 2) < slices;

        // this lane's (unit, utterance): row tile = wave, unit kq of the tile, utterance n16; its 4 accumulator registers = the 4 gates
        const int eu = slice * LX_UNITS + wave * 4 + kq, eb = b0 + n16;
        const bool e_ok = eu < hidden && eb < batch;
        const size_t gate_off = static_cast<size_t>(min(eb, batch - 1)) * (4 * hidden) + min(eu, hidden - 1);
        const bool q_ok = kq == 0 && (slice * LX_UNITS + wave * 4) < hidden && eb < batch;     // lanes 0..15 publish the tile's 4 units of one utterance
        // where this wave's 8-byte pieces go: k-step slice/2, k-octet 2*(slice&1) + (wave>>1), halves 4*(wave&1) .. +3 of the fragment
        const int pub_off = lx_publish_offset(slice, wave, n16);
        float c_state = 0.f;
        float hs, hs_inv;                                  // the utterance's power of two for h (1 unless its activations have decayed)
        lx_h_scale(ws[LX_W_RANGE + min(eb, batch - 1)], hs, hs_inv);
        const float inv_b = inv * hs_inv;

        // gate pre-activations two frames ahead (HBM / last-level cache latency is longer than a step)
        float pre_a[4], pre_b[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            pre_a[g] = gates_in[gate_off + g * hidden];
            pre_b[g] = frames > 1 ? gates_in[static_cast<size_t>(batch) * (4 * hidden) + gate_off + g * hidden] : 0.f;
        }

#if NBASR_LX_STAMPS
        const bool stamping = tile == 0 && wave == 0 && lane == 0;
        unsigned long long* const stamp_base = reinterpret_cast<unsigned long long*>(tiles + static_cast<size_t>(n_tiles) * LX_TILE_BYTES) +
                                               static_cast<size_t>(slice) * LX_STAMP_FRAMES * LX_STAMP_POINTS;
        unsigned long long st[LX_STAMP_POINTS] = {};
#endif
        int t = 0;
        bool failed = false;
        for (; t < frames; ++t) {
            LX_STAMP(0);
            xfloat4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            const int par = (t - 1) & 1;
            if (t > 0) {
                // 1. every publishing wave of the tile has set its flag to t (= it has stored h_(t-1) and L2 has the bytes)
                const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
                bool ok = false;
                for (;;) {
                    unsigned f0, f1;
                    asm volatile("global_load_dword %0, %2, off sc1\n\tglobal_load_dword %1, %2, off offset:256 sc1\n\ts_waitcnt vmcnt(0)"
                                 : "=&v"(f0), "=&v"(f1) : "v"(tflags + lane) : "memory");
                    ok = __all((!watch0 || f0 >= static_cast<unsigned>(t)) && (!watch1 || f1 >= static_cast<unsigned>(t)));
                    if (ok || __builtin_amdgcn_s_memrealtime() - t_start > LX_TIMEOUT_TICKS) break;
                    __builtin_amdgcn_s_sleep(1);
                }
                if (!ok && lane == 0) { s_role[3] = 1; __hip_atomic_store(ws + LX_W_STATUS, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                LX_STAMP(1);
                // 2. this wave's quarter of the image (k-steps 4*wave .. +3, both terms: 8 KiB) straight into LDS; sc1 = past this CU's L1
                const unsigned char* img = tile_ws + par * LX_IMAGE_BYTES;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int chunk = wave * 8 + j;        // 1 KiB = one (k-step, term) plane of 64 lanes x 16 B
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(img + chunk * 1024 + lane * 16),
                                                     (__attribute__((address_space(3))) void*)(reinterpret_cast<unsigned char*>(&htile[par][0][0][0]) + chunk * 1024),
                                                     16, 0, 16);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                LX_STAMP(2);
                __syncthreads();                           // ONE barrier per step: all four quarters are in LDS
                LX_STAMP(3);
                if (s_role[3]) { failed = true; break; }   // (workgroup-uniform: read behind the barrier)
            }
            // the gate pre-activations of frame t + 2, requested HERE: every wait above is `vmcnt(0)` (the counter is in issue order, so
            // a younger flag load or DMA cannot be waited for without them), and the matrix work below covers their trip
            float pre_c[4] = {0.f, 0.f, 0.f, 0.f};
            if (t + 2 < frames) {
                const float* gin = gates_in + static_cast<size_t>(t + 2) * batch * (4 * hidden) + gate_off;
#pragma unroll
                for (int g = 0; g < 4; ++g) pre_c[g] = gin[g * hidden];
            }
            if (t > 0) lx_row_tile_product(htile[par], wfrag, lane, acc0, acc1);       // 3. this wave's row tile x all of K
            // 4. gates, cell, h -- in the registers the sums arrived in
            const xfloat4 s = (acc0 + acc1 * 0.00048828125f) * inv_b;                // hi*hi + 2^-11 (hi*lo' + lo'*hi), then the weights' 2^-e and h's 2^-k
#if NBASR_LX_STAMPS
            { xfloat4 ss = s; asm volatile("" : "+v"(ss)); }
#endif
            LX_STAMP(4);
            const float h_new = lx_cell_update(s, pre_a, c_state);
#if NBASR_LX_STAMPS
            { float hh = h_new; asm volatile("" : "+v"(hh)); }
#endif
            LX_STAMP(5);
            // 5. h as it is consumed, 4 units x one utterance per 8-byte piece
            lx_publish(lx_split_h(h_new, hs), n16, q_ok, tile_ws + (t & 1) * LX_IMAGE_BYTES + pub_off);
            // the flag follows the bytes: drain this wave's stores (L2 has them), then one lane sets the wave's word
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) asm volatile("global_store_dword %0, %1, off" :: "v"(tflags + slice * LX_WAVES + wave), "v"(static_cast<unsigned>(t + 1)) : "memory");
            LX_STAMP(6);
            if (e_ok) h_out[(static_cast<size_t>(eb) * frames + t) * hidden + eu] = h_new;
#pragma unroll
            for (int g = 0; g < 4; ++g) { pre_a[g] = pre_b[g]; pre_b[g] = pre_c[g]; }
            LX_STAMP(7);
#if NBASR_LX_STAMPS
            if (stamping && t < LX_STAMP_FRAMES)
                for (int i = 0; i < LX_STAMP_POINTS; ++i) stamp_base[t * LX_STAMP_POINTS + i] = st[i];
#endif
        }
        if (failed) {                                      // timed out: make the failure visible in the output too
            if (e_ok) {
                const float nan = __builtin_nanf("");
                for (int u = t; u < frames; ++u) h_out[(static_cast<size_t>(eb) * frames + u) * hidden + eu] = nan;
            }
            break;
        }
        if (e_ok) cell[static_cast<size_t>(eb) * hidden + eu] = c_state;
        if (rank == 0 && threadIdx.x == 0) atomicAdd(ws + LX_W_TILES_DONE, 1u);
    }

    // ---- the last workgroup to leave checks that every tile was computed (an XCD that never collected its slices claims nothing) ----
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned before = atomicAdd(ws + LX_W_EXITS, 1u);
        if (before + 1 == static_cast<unsigned>(total_wgs)) {
            const unsigned done = atomicAdd(ws + LX_W_TILES_DONE, 0u);
            if (done != static_cast<unsigned>(n_tiles)) __hip_atomic_store(ws + LX_W_STATUS, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

}  // namespace nbasr

using namespace nbasr;

static inline int lx_slices(int hidden) { return (hidden + LX_UNITS - 1) / LX_UNITS; }

extern "C" size_t nbasr_lstm_packed_whh16_bytes(int hidden)
{
    if (hidden <= 0 || hidden > LX_KSTEPS * 32) return 0;
    return LX_PACK_HEADER_BYTES + static_cast<size_t>(lx_slices(hidden)) * LX_WAVES * LX_KSTEPS * 2 * 64 * 16;
}

extern "C" int nbasr_lstm_pack_whh16(const float* w_hh, void* packed, int hidden, nbasr_stream_t stream)
{
    clear_error();
    NBASR_REQUIRE(hidden > 0 && hidden % 4 == 0 && hidden <= LX_KSTEPS * 32, NBASR_EINVAL,
                  "nbasr_lstm_pack_whh16: hidden=%d must be a positive multiple of 4, at most %d", hidden, LX_KSTEPS * 32);
    NBASR_REQUIRE(w_hh && packed, NBASR_ENULL, "nbasr_lstm_pack_whh16: NULL pointer");
    NBASR_REQUIRE(aligned16(packed), NBASR_EALIGN, "nbasr_lstm_pack_whh16: packed must be 16-byte aligned");
    zero_async(packed, LX_PACK_HEADER_BYTES, as_stream(stream));
    hipLaunchKernelGGL(lx_absmax_kernel, dim3(256), dim3(256), 0, as_stream(stream), w_hh, static_cast<size_t>(4) * hidden * hidden,
                       static_cast<unsigned*>(packed));
    hipLaunchKernelGGL(lx_pack_kernel, dim3(512), dim3(256), 0, as_stream(stream), w_hh, static_cast<unsigned char*>(packed), hidden, lx_slices(hidden));
    return launch_status("nbasr_lstm_pack_whh16");
}

// The grid: one workgroup per compute unit of the device (every XCD then gets the workgroups of all its units under the round-robin
// dealing, i.e. all `slices` it needs -- and nothing depends on that).  A device with fewer than `slices` units cannot run the form.
static int lx_grid(int hidden)
{
    static std::mutex m;
    static int cached[64] = {};
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess || device < 0 || device >= 64) { (void)hipGetLastError(); return 0; }
    std::lock_guard<std::mutex> lock(m);
    if (cached[device] == 0) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus <= 0) { (void)hipGetLastError(); return 0; }
        cached[device] = cus;
    }
    return cached[device] >= lx_slices(hidden) ? cached[device] : 0;
}

extern "C" int nbasr_lstm_recurrence_frames16(const float* gates_ws, const void* packed_whh16, float* cell_ws, float* h_out, void* xcd_ws,
                                              int batch, int frames, int hidden, nbasr_stream_t stream)
{
    clear_error();
    NBASR_REQUIRE(batch >= 0 && frames >= 0 && hidden > 0, NBASR_EINVAL, "nbasr_lstm_recurrence_frames16: bad sizes");
    NBASR_REQUIRE(hidden % 4 == 0, NBASR_EALIGN, "nbasr_lstm_recurrence_frames16: hidden=%d must be a multiple of 4", hidden);
    if (batch == 0 || frames == 0) return NBASR_OK;
    NBASR_REQUIRE(gates_ws && packed_whh16 && cell_ws && h_out && xcd_ws, NBASR_ENULL, "nbasr_lstm_recurrence_frames16: NULL pointer");
    NBASR_REQUIRE(aligned16(packed_whh16) && aligned16(xcd_ws), NBASR_EALIGN, "nbasr_lstm_recurrence_frames16: packed_whh16, xcd_ws must be 16-byte aligned");
    NBASR_REQUIRE(nbasr_lstm_xcd_workspace_bytes(batch, hidden) != 0, NBASR_EINVAL, "nbasr_lstm_recurrence_frames16: batch=%d hidden=%d does not fit the form "
                  "(hidden <= %d, batch <= %d); use nbasr_lstm_recurrence_packed", batch, hidden, LX_KSTEPS * 32, LX_MAX_TILES * 16);
    // the images are zeroed by every call (a node of the cached graph): the rows of k beyond `hidden` and the columns beyond `batch` are
    // never written, and a NaN pattern left there by an earlier owner of the memory would turn 0 x NaN into NaN sums
    const size_t tiles_n = (batch + 15) / 16;
    struct Ctx { hipStream_t s; const float* gates; const unsigned char* w; float* cell; float* h; unsigned char* tiles; size_t tiles_bytes; int batch, frames, hidden, prio, wpb, tpb; };
    Ctx ctx{as_stream(stream), gates_ws, static_cast<const unsigned char*>(packed_whh16), cell_ws, h_out,
            static_cast<unsigned char*>(xcd_ws) + LX_HEADER_WORDS * sizeof(unsigned), tiles_n * LX_TILE_BYTES, batch, frames, hidden, 1,
            batch <= 32 ? 2 : 4, 1};
    // (tiles per workgroup, NBASR_LX_TPB: walking 2 / 4 utterance tiles with one fetch of the weights halves / quarters the bytes a frame
    // moves, but a launch then lasts 2-4 x as long and the CHAIN becomes the critical path: 10 630 -> 10 150 -> 8 690 utterances/s at 64)
    // (same-box A/B of the shapes, pipelined utterances/s: 8 utterances 5 349 / 5 381 / - with 1 / 2 / 4 waves per workgroup, 16: 7 613 / 7 685 / 6 968,
    // 64: 9 871 / 10 299 / 10 413 (8 waves: 8 800 -- a 512-thread workgroup finds no room beside the encoder's); issue priority for the
    // chain's waves 10 413 -> 10 554 at 64, within noise below)
    if (const char* force = getenv("NBASR_LX_WPB")) ctx.wpb = (force[0] == '1') ? 1 : (force[0] == '2') ? 2 : 4;      // (A/B hook; every form gives the same bits)
    if (const char* force = getenv("NBASR_LX_PRIO")) ctx.prio = atoi(force);
    if (const char* force = getenv("NBASR_LX_TPB")) ctx.tpb = std::min(std::max(atoi(force), 1), 4);
    const ChainKey key{{gates_ws, packed_whh16, cell_ws, h_out, xcd_ws}, {batch, frames, hidden, 16, ctx.wpb * 8 + ctx.tpb}};
    return replay_chain(ctx.s, key, "nbasr_lstm_recurrence_frames16", [](void* p) {
        const Ctx& c = *static_cast<const Ctx*>(p);
        unsigned* const range = reinterpret_cast<unsigned*>(c.tiles) - (LX_HEADER_WORDS - LX_W_RANGE);
        zero_async(range, (LX_HEADER_WORDS - LX_W_RANGE) * sizeof(unsigned) + c.tiles_bytes, c.s);                     // the range words + the images
        hipLaunchKernelGGL(lx_gate_range_kernel, dim3((c.frames + 7) / 8, c.batch), dim3(256), 0, c.s, c.gates, range, c.batch, c.frames, 4 * c.hidden);
        const int row_tiles = (c.hidden + 3) / 4;
        const int n_tiles = (c.batch + 15) / 16;
        const dim3 grid((row_tiles + c.wpb - 1) / c.wpb, (n_tiles + c.tpb - 1) / c.tpb);
        for (int t = 0; t < c.frames; ++t) {
            if (c.wpb == 1) hipLaunchKernelGGL(lstm_step16_kernel<1>, grid, dim3(64), 0, c.s, c.gates, c.w, c.cell, c.h, c.tiles, c.batch, c.frames, c.hidden, t, c.prio, c.tpb);
            else if (c.wpb == 2) hipLaunchKernelGGL(lstm_step16_kernel<2>, grid, dim3(128), 0, c.s, c.gates, c.w, c.cell, c.h, c.tiles, c.batch, c.frames, c.hidden, t, c.prio, c.tpb);
            else hipLaunchKernelGGL(lstm_step16_kernel<4>, grid, dim3(256), 0, c.s, c.gates, c.w, c.cell, c.h, c.tiles, c.batch, c.frames, c.hidden, t, c.prio, c.tpb);
        }
    }, &ctx);
}

extern "C" size_t nbasr_lstm_xcd_workspace_bytes(int batch, int hidden)
{
    if (batch <= 0 || hidden <= 0 || hidden % 4 || hidden > LX_KSTEPS * 32) return 0;
    const size_t tiles = (batch + 15) / 16;
    if (tiles > LX_MAX_TILES) return 0;
    return LX_HEADER_WORDS * sizeof(unsigned) + tiles * LX_TILE_BYTES + LX_STAMP_BYTES;
}

extern "C" int nbasr_lstm_recurrence_xcd(const float* gates_ws, const void* packed_whh16, float* cell_ws, float* h_out, void* xcd_ws,
                                         int batch, int frames, int hidden, int flags, nbasr_stream_t stream)
{
    clear_error();
    NBASR_REQUIRE(batch >= 0 && frames >= 0 && hidden > 0, NBASR_EINVAL, "nbasr_lstm_recurrence_xcd: bad sizes");
    NBASR_REQUIRE(hidden % 4 == 0, NBASR_EALIGN, "nbasr_lstm_recurrence_xcd: hidden=%d must be a multiple of 4", hidden);
    if (batch == 0 || frames == 0) return NBASR_OK;
    NBASR_REQUIRE(gates_ws && packed_whh16 && cell_ws && h_out && xcd_ws, NBASR_ENULL, "nbasr_lstm_recurrence_xcd: NULL pointer");
    NBASR_REQUIRE(aligned16(packed_whh16) && aligned16(xcd_ws), NBASR_EALIGN, "nbasr_lstm_recurrence_xcd: packed_whh16, xcd_ws must be 16-byte aligned");
    const size_t ws_bytes = nbasr_lstm_xcd_workspace_bytes(batch, hidden);
    NBASR_REQUIRE(ws_bytes != 0, NBASR_EINVAL, "nbasr_lstm_recurrence_xcd: batch=%d hidden=%d does not fit the form (hidden <= %d, batch <= %d); "
                  "use nbasr_lstm_recurrence_packed", batch, hidden, LX_KSTEPS * 32, LX_MAX_TILES * 16);
    const int grid = lx_grid(hidden);
    NBASR_REQUIRE(grid > 0, NBASR_EINVAL, "nbasr_lstm_recurrence_xcd: the device has fewer compute units than the %d slices of hidden=%d",
                  lx_slices(hidden), hidden);
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    hipError_t e = hipStreamIsCapturing(as_stream(stream), &capturing);
    if (e != hipSuccess) { (void)hipGetLastError(); capturing = hipStreamCaptureStatusNone; }
    const bool captured = capturing != hipStreamCaptureStatusNone;
    int device = 0;
    e = hipGetDevice(&device);
    NBASR_REQUIRE(e == hipSuccess && device >= 0 && device < 64, NBASR_EINVAL, "nbasr_lstm_recurrence_xcd: device %d out of range", device);
    // Two of these grids can sit on the chip together; a third one could split an XCD's units with them so that none of the three
    // collects its slices (they would time out, bounded, and say so).  Launches from different streams of the process are therefore
    // chained, stream-ordered (each waits for the event behind the previous one; no host synchronisation).  A stream under capture
    // cannot take part in that chain: a captured launch (memset + kernel nodes) is ordered by its graph alone, and keeping replays of
    // such a graph from overlapping with other resident recurrences is the caller's business (the executor replays a graph on the
    // stream its other forwards run on).
    static std::mutex chain_mutex;
    static hipEvent_t chain_done[64] = {};
    std::unique_lock<std::mutex> lock(chain_mutex);
    hipEvent_t& done = chain_done[device];
    e = hipSuccess;
    if (!captured) {
        if (done == nullptr) e = hipEventCreateWithFlags(&done, hipEventDisableTiming);
        else e = hipStreamWaitEvent(as_stream(stream), done, 0);
    }
    if (e != hipSuccess) { set_error("nbasr_lstm_recurrence_xcd: %s", hipGetErrorString(e)); return static_cast<int>(e); }
    zero_async(xcd_ws, ws_bytes, as_stream(stream));
    const int n_tiles = (batch + 15) / 16;
    hipLaunchKernelGGL(lx_gate_range_kernel, dim3((frames + 7) / 8, batch), dim3(256), 0, as_stream(stream), gates_ws,
                       static_cast<unsigned*>(xcd_ws) + LX_W_RANGE, batch, frames, 4 * hidden);
    hipLaunchKernelGGL(lstm_xcd_kernel, dim3(grid), dim3(64 * LX_WAVES), 0, as_stream(stream), gates_ws, static_cast<const unsigned char*>(packed_whh16),
                       cell_ws, h_out, static_cast<unsigned*>(xcd_ws), batch, frames, hidden, lx_slices(hidden), n_tiles, grid, flags);
    const int rc = launch_status("nbasr_lstm_recurrence_xcd");
    if (rc != NBASR_OK || captured) return rc;
    e = hipEventRecord(done, as_stream(stream));
    if (e != hipSuccess) { set_error("nbasr_lstm_recurrence_xcd: hipEventRecord: %s", hipGetErrorString(e)); return static_cast<int>(e); }
    return NBASR_OK;
}
