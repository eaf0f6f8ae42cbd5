// Dense k=8 PadConvRelu (reference model.py:82-89, ops.py:24-30) as an fp32-ACCURATE implicit GEMM on the bf16
// matrix cores of gfx950 (v_mfma_f32_16x16x32_bf16, 16x the rate of the fp32 MFMA; the 16x16x32 shape holds a higher clock
// than 32x32x16 under the chip's power management: +7-11 % measured on these layers at equal cycles per flop).
//
// Every fp32 operand is split EXACTLY into three bf16 terms  v = hi + mid + lo  (hi = rne(v), mid = rne(v - hi),
// lo = rne(v - hi - mid): 3 x 8 significand bits cover fp32's 24) and the product is evaluated as the six terms
//     a*b ~= ah*bh + (ah*bm + am*bh) + (ah*bl + al*bh + am*bm)
// The dropped terms (am*bl, al*bm, al*bl) are <= 2^-24 |a*b|, i.e. below the rounding error of an fp32 multiply;
// bf16 x bf16 products are exact in fp32 and the MFMA accumulates in fp32, so the result carries fp32-level error
// (checked against an fp64 oracle in tests/).  6 MFMAs at 16x the rate = 2.67x the fp32 matrix peak.
//
// Second scheme, HALF the MFMAs (policy SplitF16x2): fp16 carries 11 significand bits, so TWO terms suffice,
//     v = hi + lo,   hi = rne16(v),   lo = rne16(v - hi)      |v - hi - lo| <= 2^-24 |v|  (23 bits + the sign of lo)
//     a*b ~= ah*bh + ah*bl + al*bh            dropped: al*bl <= 2^-24 |a*b|
// i.e. 3 MFMAs per fp32 product (5.3x the fp32 matrix peak).  fp16's RANGE (6e-5 .. 65504 for normal numbers) is the
// catch, and the model's activations really do leave it (with the reference's default initialisation they shrink to 1e-9).
// Both operands are therefore brought into range by EXACT power-of-two scalings that the epilogue undoes:
//   * x of utterance b is multiplied by 2^kx[b] so that its largest magnitude lands in [2^14, 2^15); the caller passes
//     absmax[b] >= max|x[b]| (nbasr_layernorm_channels produces it for free while writing x; the model input is
//     ranged by nbasr_input_range);
//   * row co of w is multiplied by 2^kw[co] at pack time (largest magnitude of the row -> [2^13, 2^14)).
// Round 3: lo is stored UNSCALED (rounds 1-2 stored lo * 2^11 to keep it a normal fp16 number and accumulated the cross terms in
// a second register set).  fp16 subnormals are honoured by v_cvt_f16_f32 and by the MFMA (tools/ubench/mfma_f16_denorm.hip), so
// lo is exact down to 2^-24: elements within 2^-16 of their utterance's / row's maximum keep all 24 bits, smaller ones are off by
// at most 2^-25 absolute = 2^-39 of that maximum -- far below fp32 resolution of any sum they enter.  What it buys: all three
// products go to ONE accumulator, which frees the second register set for a BLOCKED sum (next paragraph).
//
// Accumulation (round 3).  Rounds 1-2 kept one running sum per output over the whole K = c_in * 8 (one rounding
// per 32-k MFMA, 150-300 roundings at the magnitude of the growing sum).  Measured per layer that chain -- not the split -- was
// the path's distance to the CPU's blocked sums: the split's representation error is 0.33 of the CPU conv's own error against
// fp64, the chain was 1.4 x (profiles/r02_layer_noise.txt: 0.90 -> 1.25 x the CPU's error across conv 1).  Now every channel
// group (16 channels x 8 taps = 4 k-blocks = 12 MFMAs) is summed from ZERO in `acc` and added to `tot` once: 12 + K/128
// roundings instead of K/32 at full magnitude, the error model of a two-level blocked sum (emulated on the CPU in
// tools/split_accumulation_model.py: 0.61 x the CPU conv's error where the single chain gave 1.0-1.4 x).
//
// GEMM view per utterance: M = c_out, N = output frames, K = (c_in, tap).  The 32 k of one MFMA are 16 input CHANNELS of
// TWO consecutive taps (a lane holds 8 consecutive k = 8 channels of one tap):
//     D[co][t] += sum_{ci<16} W[co][g*16+ci][tap] * x[g*16+ci][t*stride + tap - lpad]      for every (group g, tap)
//  * weights are split and re-laid-out ONCE (nbasr_pack_dense_weights) into the exact LDS image of each
//    (row tile, channel group, tap quad): [split][tap][ci half][128 rows][8 ci] 16-bit = 32 KiB (fp16 x 2) or 48 KiB
//    (bf16 x 3), so a K-step's weights are a straight copy done by LDS-DMA (global_load_lds_dwordx4, no VGPRs), double-buffered;
//  * the input tile of a channel group is fetched as aligned 4-frame quads, split on the fly and stored TRANSPOSED
//    [ci half][frame][8 ci] so a B fragment (8 consecutive channels of one frame) is one aligned, bank-conflict-free
//    ds_read_b128; it is staged once per group and reused by all 8 taps (sliding window resolved by the row index;
//    stride-2 rows are de-interleaved by parity so the 16 lanes of a fragment read hit consecutive rows);
//  * one 512-thread workgroup per CU: 128 x 256 tile / 8 waves (2 x 4, 64 x 64 each = 4 x 4 MFMA tiles), K-step =
//    16 channels x 4 taps (2 for bf16 x 3 at stride 2: LDS budget), one barrier per step; the two halves of the workgroup run
//    a step in opposite order (stage-then-multiply / multiply-then-stage) so a wave's memory phase sits beside its SIMD
//    partner's MFMAs, and the multiply-first half runs at s_setprio 1 so that it really finishes first;
//  * accumulation is a two-level blocked sum (per channel group, then the total): see the scheme notes above.
#include "storage.h"

#include <cstdlib>

#include <type_traits>

namespace nbasr {

typedef float floatx4 __attribute__((ext_vector_type(4)));

constexpr int PB_CI = 16, PB_TAPS = 8;               // channels per group, conv taps
// frames per tile = 64 * NJ: NJ = 4 (256 frames; every kernel) or 2 (128 frames; the image-path fp16 kernel only, where a small batch
// leaves most compute units without a workgroup -- twice the workgroups, half the matrix work per K-step: VERDICT r4 next 5b).  A wave owns
// 16 MI rows x 16 NJ frames; the K order of every output is the same whatever the tile, so results are bit-identical.
constexpr int pb_frames(int nj) { return 64 * nj; }
// rows per tile = 32 * MI: MI = 4 (128 rows; every kernel) or 5 (160 rows; the image-path fp16 kernel only, for layers whose
// 128-row tiling leaves a mostly empty last row tile or a partial last round of workgroups -- C_out = 800 and 1200)
constexpr int pb_rows(int mi) { return 32 * mi; }
constexpr int PB_THREADS = 512;                                              // 8 waves: 2 (rows) x 4 (frames), 64 x 64 each

// ---- the two operand-splitting schemes ---------------------------------------------------------------------------
struct SplitBf16x3 {
    static constexpr int NS = 3;
    typedef __bf16 vec8 __attribute__((ext_vector_type(8)));
    static constexpr bool SCALED = false;                        // bf16 has fp32's exponent range
    __host__ __device__ static constexpr int taps_per_step(int stride) { return stride == 1 ? 4 : 2; }   // LDS budget: 160 KiB
    __device__ static __forceinline__ void split(float v, unsigned short (&s)[NS]) {
        const __bf16 hi = static_cast<__bf16>(v);
        const float r1 = v - static_cast<float>(hi);
        const __bf16 mid = static_cast<__bf16>(r1);
        const __bf16 lo = static_cast<__bf16>(r1 - static_cast<float>(mid));
        s[0] = __builtin_bit_cast(unsigned short, hi);
        s[1] = __builtin_bit_cast(unsigned short, mid);
        s[2] = __builtin_bit_cast(unsigned short, lo);
    }
    __device__ static __forceinline__ floatx4 mfma(vec8 a, vec8 b, floatx4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
    static constexpr const char* NAME = "nbasr_dense_conv1d_packed(bf16x3)";
};

struct SplitF16x2 {
    static constexpr int NS = 2;
    typedef _Float16 vec8 __attribute__((ext_vector_type(8)));
    static constexpr bool SCALED = true;                         // operands are range-normalised by powers of two
    __host__ __device__ static constexpr int taps_per_step(int) { return 4; }
    __device__ static __forceinline__ void split(float v, unsigned short (&s)[NS]) {
        const _Float16 hi = static_cast<_Float16>(v);
        const _Float16 lo = static_cast<_Float16>(v - static_cast<float>(hi));      // unscaled; subnormal below 2^-14 (exact to 2^-24)
        s[0] = __builtin_bit_cast(unsigned short, hi);
        s[1] = __builtin_bit_cast(unsigned short, lo);
    }
    __device__ static __forceinline__ floatx4 mfma(vec8 a, vec8 b, floatx4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
    static constexpr const char* NAME = "nbasr_dense_conv1d_packed(f16x2)";
};

// bf16 path (BASELINE config 4): operands ARE bfloat16 -- one term, one MFMA per product, no scaling (bf16 has fp32's exponent
// range).  Image path only: the producer (LayerNorm, or the re-layout of the model input) writes the operand image; the
// result is rounded once to bf16 and leaves through an LDS transpose as whole 128-byte row segments.
struct PlainBf16 {
    static constexpr int NS = 1;
    typedef __bf16 vec8 __attribute__((ext_vector_type(8)));
    static constexpr bool SCALED = false;
    __host__ __device__ static constexpr int taps_per_step(int) { return 8; }      // all 8 taps of a channel group per K-step
    __device__ static __forceinline__ void split(float v, unsigned short (&s)[NS]) {
        s[0] = __builtin_bit_cast(unsigned short, static_cast<__bf16>(v));
    }
    __device__ static __forceinline__ floatx4 mfma(vec8 a, vec8 b, floatx4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
    static constexpr const char* NAME = "nbasr_dense_conv1d_packed(bf16)";
};

template <class P> constexpr bool has_image_path() { return P::SCALED || P::NS == 1; }

template <class P> constexpr size_t pb_group_bytes(int mi) { return static_cast<size_t>(P::NS) * PB_TAPS * pb_rows(mi) * PB_CI * 2; }   // packed weights of one (row tile, channel group)

template <class P, int S, int MI = 4, int NJ = 4>
struct GeoP {
    static constexpr int PBM = pb_rows(MI);                      // rows per tile; a wave owns 16 * MI of them
    static constexpr int PBN = pb_frames(NJ);                    // frames per tile; a wave owns 16 * NJ of them
    static constexpr int TP = P::taps_per_step(S);               // taps per K-step
    static constexpr int QSTEPS = PB_TAPS / TP;                  // K-steps per channel group
    static constexpr int A_STEP_BYTES = P::NS * TP * PBM * PB_CI * 2;
    static constexpr int XR = (PBN - 1) * S + PB_TAPS;           // input frames needed per channel
    static constexpr int XRH = (XR + 1) / 2;                     // rows per parity plane (stride 2)
    static constexpr int ROWS = (S == 1) ? XR : 2 * XRH;         // rows per (split, channel half) plane that hold data
    // Round 6: the planes are PITCHED to whole 16-row units.  A B fragment read (ds_read_b128) is served in groups of 16 lanes that mix two
    // k quarters -- lanes {0-3, 12-15} of channel half 0 with lanes {20-27} of half 1, one plane further -- and the group is conflict-free
    // only if a plane's pitch is a multiple of 256 bytes: its 16 rows x 16 bytes then tile all 64 banks.  At 263 (stride 1) / 518
    // (stride 2) rows the second half landed 28 / 24 banks off and every fragment read of the input tile took two LDS cycles per group
    // (profiles/r05_pmc_lds_cfg3_bf16.csv: 0.31-0.34 of the LDS-active cycles were bank conflicts).
    static constexpr int RP = (ROWS + 15) / 16 * 16;             // plane pitch in rows
    static constexpr int X_BYTES = P::NS * RP * PB_CI * 2;
    // the tile is fetched as ALIGNED 4-frame quads (one global_load_dwordx4 per channel): NQUADS covers XR rows at any
    // misalignment of the tile's first frame; an item = (quad, channel pair), 8 consecutive quads x 8 pairs per wave
    static constexpr int NQUADS = ((XR + 3 + 3) / 4 + 7) / 8 * 8;
    static constexpr int XITEMS = NQUADS * (PB_CI / 2);
    static constexpr int NCHUNK = QSTEPS > 1 ? QSTEPS - 1 : 1;   // the next group's tile is staged in QSTEPS-1 chunks (register staging path)
    static constexpr int XI = (XITEMS + PB_THREADS * NCHUNK - 1) / (PB_THREADS * NCHUNK);   // items per thread per chunk
    // operand slots: two (the transfers of step s + 1 are issued in step s).  Round 6 also built THREE for the one-term bf16 flavour
    // (step s issues the operands of step s + 2, a bare s_barrier per step; its tests passed): 0.983 against 0.988 ms of dense convs per
    // forward at 32 x 1600 -- nothing, as round 5 found for the fp16 x 2 flavour.  What bounds that flavour is LDS READ bandwidth: a wave's
    // 80 x 64 register tile re-uses an A fragment 4 times and a B fragment 5 times, 0.45 KiB of ds_read_b128 per 16-cycle MFMA and wave =
    // 230 of the CU's 256 bytes per clock at the full matrix rate (the split flavours issue 2-3 MFMAs per fragment pair).
    static constexpr int RING = 2;
    static constexpr int OPERAND_BYTES = RING * (A_STEP_BYTES + X_BYTES);
    static constexpr int STAGING_BYTES = 8 * 32 * 68 * 4;            // the fp32 epilogue's output staging (8 waves x 32 rows x 68 floats)
    static constexpr int LDS_BYTES = OPERAND_BYTES > STAGING_BYTES ? OPERAND_BYTES : STAGING_BYTES;
    __device__ static constexpr int rowmap(int row) { return (S == 1) ? row : (row & 1) * XRH + (row >> 1); }
};

struct PackedConvArgs {
    const float* x; const unsigned char* wp; const float* bias;
    const float* s0; const float* s1; const float* s2;
    float* y;
    int c_in, frames_in, ld_in, c_out, frames_out, ld_out, lpad;
    int n_groups, n_mt, n_nt, batch;
    LnRef ln_x;                  // pending LayerNorm of the input (deferred normalisation, nbasr.h)
    const float* x_absmax;       // SCALED schemes: (batch) upper bounds of max|x[b]|
    const float* w_inv_scale;    // SCALED schemes: (n_mt * 128) 2^-kw[co], tail of the packed buffer
    int x_is_image;              // x points to the pre-split fp16 image (XIMG kernels)
    // per-utterance routing on the device (the model input, nbasr_input_range): range[4 b] = {max|x|, quietest frame's max,
    // non-finite flag, -}; a workgroup whose utterance is (not) extreme exits at once when sel_want is 0 (1); -1: no routing
    const float* x_range;
    int sel_want;
    int staged_epilogue;         // fp32 output through LDS in whole row segments (always 1; the direct stores were 1.2 % slower end to end)
    float* part;                 // fp32 output, no skips: also emit the partial LayerNorm statistics of y -- per frame the (mean, M2) over every
                                 // 16 channels -- to part[16-channel unit][batch][2][ld_out] (merged by stats_finalize_kernel); NULL: none
};

// an utterance the scaled fp16 scheme must not take: non-finite samples, or a frame > 2^12 below the loudest sample (the unscaled
// lo term is exact to 2^-24 with the maximum at 2^14..2^15: a frame 2^12 below it still has its top three binades at full precision
// and everything below within 2^-27 of the frame's own scale; rounds 1-2, with lo pre-scaled by 2^11, drew the line at 2^20)
__device__ __forceinline__ bool range_is_extreme(const float* r) { return r[2] != 0.f || r[1] < r[0] * 2.44140625e-04f; }

// 2^k that moves a magnitude with biased exponent field e to [2^target, 2^(target+1)), and its inverse; (1, 1) for zero
__host__ __device__ inline void pow2_normaliser(float absmax, int target, float& scale, float& inv)
{
    unsigned bits;
    __builtin_memcpy(&bits, &absmax, 4);
    const int e = static_cast<int>((bits >> 23) & 0xffu);
    int k = (bits & 0x7fffffffu) ? (127 + target) - e : 0;
    k = k > 126 ? 126 : (k < -126 ? -126 : k);
    const unsigned sb = static_cast<unsigned>(127 + k) << 23, ib = static_cast<unsigned>(127 - k) << 23;
    __builtin_memcpy(&scale, &sb, 4);
    __builtin_memcpy(&inv, &ib, 4);
}

__device__ __forceinline__ unsigned pack2(unsigned short a, unsigned short b) {
    return static_cast<unsigned>(a) | (static_cast<unsigned>(b) << 16);
}

// ---- one-time weight split + re-layout -------------------------------------------------------------------------
// packed element (mt, g, q, split, tp, half = ci_l/8, co_l, ci_l%8) <- W[mt*PB_M + co_l][g*16 + ci_l][q*TP + tp]  (zero outside);
// TP = taps per K-step of the kernel that will consume the image
// SCALED schemes: scales[co] = 2^kw[co], scales[rows + co] = 2^-kw[co]   (rows = n_mt * PB_M; rows beyond c_out: 1)
__global__ __launch_bounds__(256) void weight_row_scales_kernel(const float* __restrict__ w, float* __restrict__ scales,
                                                                int c_out, int row_elems, int rows)
{
    __shared__ float s_max[4];
    const int co = blockIdx.x;
    float m = 0.f;
    if (co < c_out)
        for (int i = threadIdx.x; i < row_elems; i += 256) m = fmaxf(m, fabsf(w[static_cast<size_t>(co) * row_elems + i]));
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
    if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        float scale, inv;
        pow2_normaliser(fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3])), 13, scale, inv);
        scales[co] = scale;
        scales[rows + co] = inv;
    }
}

template <class P>
__global__ __launch_bounds__(256) void pack_dense_weights_kernel(const float* __restrict__ w, unsigned short* __restrict__ wp,
                                                                 const float* __restrict__ row_scale,
                                                                 int c_out, int c_in, int n_mt, int n_groups, int PB_TP, int PB_M)
{
    const int PB_QSTEPS = PB_TAPS / PB_TP;
    const long long total = static_cast<long long>(n_mt) * n_groups * PB_QSTEPS * PB_TP * PB_M * PB_CI;
    for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < total;
         i += static_cast<long long>(gridDim.x) * blockDim.x) {
        long long e = i;
        const int ci_l = e % PB_CI; e /= PB_CI;
        const int co_l = e % PB_M; e /= PB_M;
        const int tp = e % PB_TP; e /= PB_TP;
        const int q = e % PB_QSTEPS; e /= PB_QSTEPS;
        const int g = e % n_groups; e /= n_groups;
        const int mt = static_cast<int>(e);
        const int co = mt * PB_M + co_l, ci = g * PB_CI + ci_l, tap = q * PB_TP + tp;
        float v = (co < c_out && ci < c_in) ? w[(static_cast<size_t>(co) * c_in + ci) * PB_TAPS + tap] : 0.f;
        if (P::SCALED) v *= row_scale[co];
        unsigned short s[P::NS];
        P::split(v, s);
        const size_t step = (static_cast<size_t>(mt) * n_groups + g) * PB_QSTEPS + q;
#pragma unroll
        for (int k = 0; k < P::NS; ++k)
            wp[((((step * P::NS + k) * PB_TP + tp) * 2 + (ci_l >> 3)) * PB_M + co_l) * 8 + (ci_l & 7)] = s[k];
    }
}

// ---- the GEMM -----------------------------------------------------------------------------------------------------
// XIMG: x is not the fp32 activation but its pre-split fp16 image written by the normalise-and-split LayerNorm kernel
// (layernorm.hip): [b][16-channel group][split][8-channel half][1 + ld_in rows][8 ch], row 0 all zero, frame t at row t + 1,
// already scaled by 2^kx[b].  The input tile then needs no vector work at all: it is gathered by LDS-DMA like the weights.
template <class P, int S, bool LNX, bool XIMG = false, int MI = 4, int NJ = 4>
__global__ __launch_bounds__(PB_THREADS, 2) void gemm_conv_split_kernel(const PackedConvArgs a)
{
    static_assert(NJ == 4 || (NJ == 2 && XIMG && P::NS == 2) || (NJ == 8 && XIMG && P::NS == 1),
                  "128-frame tiles exist for the fp16 image path only, 512-frame tiles for the one-term bf16 flavour only");
    static_assert(!(XIMG && LNX) && !(XIMG && !has_image_path<P>()), "the image path: scaled fp16 scheme or plain bf16, no LayerNorm on load");
    static_assert(XIMG || P::NS > 1, "plain bf16 operands exist as an image only");
    static_assert(MI == 4 || ((MI == 5 || MI == 3 || MI == 2) && XIMG), "160-, 96- and 64-row tiles exist for the image path only");
    using G = GeoP<P, S, MI, NJ>;
    constexpr int PB_M = G::PBM, WROWS = 16 * MI;                // rows per tile, rows per wave
    constexpr int PB_N = G::PBN, WCOLS = 16 * NJ;                // frames per tile, frames per wave
    using vec8 = typename P::vec8;
    constexpr int TP = G::TP, QS = G::QSTEPS, ASTEP = G::A_STEP_BYTES;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const Abuf = smem;                       // [RING][ASTEP]    weights of the current / next (/ next but one) K-step
    unsigned char* const Xbase = smem + G::RING * ASTEP;    // [RING][X_BYTES]  input tile of the current / next (/ next but one) channel group

    // XCD-aware tile order (the bijective remap of gemm_conv.hip), frame-tile major
    const int nwg = gridDim.x, id = blockIdx.x;
    const int xq = nwg >> 3, xr = nwg & 7, xcd = id & 7;
    const int L = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (id >> 3);
    int mt_i, b, nt_i;
    {                                // frame-tile major: all row tiles of a frame tile next to each other, they share its operand image through L2
        mt_i = L % a.n_mt;
        const int rest = L / a.n_mt;
        b = rest / a.n_nt;
        nt_i = rest - b * a.n_nt;
    }
    const int m0 = mt_i * PB_M, n0 = nt_i * PB_N;
    if (a.x_range && a.sel_want >= 0 && static_cast<int>(range_is_extreme(a.x_range + 4 * b)) != a.sel_want) return;   // whole workgroup

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    // Waves 0-3 and 4-7 share the four SIMDs pairwise and run the SAME K-step between two barriers.  Measured with
    // s_memtime stamps: issuing the step's vector-memory work (6 x 1 KiB LDS-DMA pieces cost ~250 cycles each to issue,
    // the tile gathers likewise) at the top of the step in every wave left the matrix pipe idle for ~1500 of ~9500 cycles.
    // So the two halves run the step in OPPOSITE order: the older wave stages first and multiplies last, the younger one
    // multiplies first and stages last -- each wave's memory phase sits beside its partner's MFMAs.
    const bool older = wave < 4;

    const float* __restrict__ xb = a.x + static_cast<size_t>(b) * a.c_in * a.ld_in;
    const int tin0 = n0 * S - a.lpad;
    const float* __restrict__ xstats = LNX ? a.ln_x.stats + static_cast<size_t>(b) * 2 * a.ld_in : nullptr;
    const unsigned char* __restrict__ wtile = a.wp + static_cast<size_t>(mt_i) * a.n_groups * pb_group_bytes<P>(MI);

    // a wave whose whole tile is out of range issues no MFMAs
    const bool wave_active = (m0 + wm * WROWS) < a.c_out && (n0 + wn * WCOLS) < a.ld_out;

    // two accumulator sets, a two-level BLOCKED sum: `acc` takes every product of ONE channel group (smallest terms first) starting
    // from zero, `tot` takes acc once per group (flush_group).  One-term operands (NS == 1) use `acc` alone.
    floatx4 acc[MI][NJ], tot[MI][NJ];                 // 16 x 16 tiles: row (lane >> 4) * 4 + r, column lane & 15
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) { acc[i][j][r] = 0.f; tot[i][j][r] = 0.f; }
    auto flush_group = [&]() {
        if constexpr (P::NS > 1) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { tot[i][j][r] += acc[i][j][r]; acc[i][j][r] = 0.f; }
        }
    };

    // ---- staging helpers ---------------------------------------------------------------------------------------
    // weights of K-step `step` (index within this row tile) -> Abuf[buf] by LDS-DMA, ASTEP/1 KiB wave copies
    auto dma_weights = [&](int step, int buf) {
        const unsigned char* src = wtile + static_cast<size_t>(step) * ASTEP;
#pragma unroll
        for (int j = 0; j < ASTEP / 1024 / 8; ++j) {
            const int chunk = wave * (ASTEP / 1024 / 8) + j;
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(src + chunk * 1024 + lane * 16),
                (__attribute__((address_space(3))) void*)(Abuf + buf * ASTEP + chunk * 1024),
                16, 0, 0);
        }
    };
    // XIMG: tile of channel group g -> X[xbuf], pieces of 1 KiB (64 lanes x one 16-byte image row each); piece i of a group
    // is issued in K-step i % QS by wave (i / QS) % 8.  LDS slot (split, half, row rr) <- image row of frame
    // tin0 + rr (stride 1) or tin0 + 2 (rr % XRH) + rr / XRH (stride 2: parity planes); frames outside [0, ld_in) -> zero row 0
    auto dma_x = [&](int g, int xbuf, int q) {
        constexpr int NP = (G::X_BYTES + 1023) / 1024;
        const unsigned char* img = reinterpret_cast<const unsigned char*>(a.x) +
                                   (static_cast<size_t>(b) * a.n_groups + g) * (2 * P::NS) * static_cast<size_t>(a.ld_in + 1) * 16;
#pragma unroll 1
        for (int i = wave * QS + q; i < NP; i += 8 * QS) {
            const int o = i * 1024 + lane * 16;
            if (o < G::X_BYTES) {
                const int plane = o / (G::RP * 16), rr = (o - plane * (G::RP * 16)) >> 4;
                const int frame = tin0 + (S == 1 ? rr : 2 * (rr % G::XRH) + rr / G::XRH);
                const int row = (rr < G::ROWS && frame >= 0 && frame < a.ld_in) ? frame + 1 : 0;     // (pitch rows: the zero row)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(img + (static_cast<size_t>(plane) * (a.ld_in + 1) + row) * 16),
                    (__attribute__((address_space(3))) void*)(Xbase + xbuf * G::X_BYTES + i * 1024), 16, 0, 0);
            }
        }
    };
    floatx4 xreg[G::XI][2];
    // item e of a chunk: quad (e >> 6) * 8 + (e & 7) (frames a0 + 4 * quad .. + 3, a0 = tin0 rounded down to a multiple
    // of 4), channel pair (e >> 3) & 7.  Quads are aligned, so each is wholly inside [0, ld_in) or wholly outside; the
    // pitch columns frames_in..ld_in-1 are zero by the layout contract (nbasr.h) and their rstd is 0.
    const int a0 = tin0 & ~3, xoff = tin0 - a0;
    float x_scale = 1.f, x_inv = 1.f;                 // SCALED schemes: 2^kx[b] and its inverse (uniform over the workgroup)
    if constexpr (P::SCALED) pow2_normaliser(a.x_range ? a.x_range[4 * b] : a.x_absmax[b], 14, x_scale, x_inv);
    auto load_x = [&](int g, int c) {
#pragma unroll
        for (int i = 0; i < G::XI; ++i) {
            const int e = tid + PB_THREADS * (c * G::XI + i);
            const int p = (e >> 3) & 7;
            const int t = a0 + 4 * ((e >> 6) * 8 + (e & 7));
            const int ci = g * PB_CI + 2 * p;
            const bool ok = e < G::XITEMS && t >= 0 && t < a.ld_in;
            floatx4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = {0.f, 0.f, 0.f, 0.f};
            if (ok && ci < a.c_in) v0 = *reinterpret_cast<const floatx4*>(xb + static_cast<size_t>(ci) * a.ld_in + t);
            if (ok && ci + 1 < a.c_in) v1 = *reinterpret_cast<const floatx4*>(xb + static_cast<size_t>(ci + 1) * a.ld_in + t);
            if (LNX && ok) {
                const floatx4 mean = *reinterpret_cast<const floatx4*>(xstats + t);
                const floatx4 rstd = *reinterpret_cast<const floatx4*>(xstats + a.ld_in + t);
                if (ci < a.c_in) {
                    const float gm = a.ln_x.gamma[ci], bt = a.ln_x.beta[ci];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v0[j] = ln_apply(v0[j], mean[j], rstd[j], gm, bt);
                }
                if (ci + 1 < a.c_in) {
                    const float gm = a.ln_x.gamma[ci + 1], bt = a.ln_x.beta[ci + 1];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v1[j] = ln_apply(v1[j], mean[j], rstd[j], gm, bt);
                }
            }
            if constexpr (P::SCALED) { v0 *= x_scale; v1 *= x_scale; }
            xreg[i][0] = v0;
            xreg[i][1] = v1;
        }
    };
    auto commit_x = [&](int xbuf, int c) {
        unsigned* const X = reinterpret_cast<unsigned*>(Xbase + xbuf * G::X_BYTES);
#pragma unroll
        for (int i = 0; i < G::XI; ++i) {
            const int e = tid + PB_THREADS * (c * G::XI + i);
            const int p = (e >> 3) & 7;
            const int row0 = 4 * ((e >> 6) * 8 + (e & 7)) - xoff;
            if (e >= G::XITEMS) continue;
            // image [split][half][row][8 ci]: channel pair p sits in half p >> 2, dword p & 3 of the 16-byte row
            unsigned* const col = X + (p >> 2) * G::RP * 4 + (p & 3);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = row0 + j;
                if (row < 0 || row >= G::XR) continue;
                unsigned short s0[P::NS], s1[P::NS];
                P::split(xreg[i][0][j], s0);
                P::split(xreg[i][1][j], s1);
                unsigned* dst = col + G::rowmap(row) * 4;
#pragma unroll
                for (int k = 0; k < P::NS; ++k) dst[k * 2 * G::RP * 4] = pack2(s0[k], s1[k]);
            }
        }
    };

    // per-lane fragment bases (bytes).  v_mfma_f32_16x16x32_bf16: lane l supplies row/column l & 15 and k = 8 * (l >> 4) ..
    // + 7; the 32 k of one MFMA are the 16 channels of TWO consecutive taps: k quarter kq = l >> 4 -> tap kq >> 1 of the
    // pair, channel half kq & 1.  Both LDS images are [..][tap][half][row][8 channels] with 16-byte rows, so the 16 lanes
    // of a quarter read 256 contiguous bytes (all 64 banks) and a fragment is one ds_read_b128.
    const int l15 = lane & 15, kq = lane >> 4;
    const int a_lane = (((kq >> 1) * 2 + (kq & 1)) * PB_M + wm * WROWS + l15) * 16;
    const int x_lane = (kq & 1) * G::RP * 16;

    // PIPE (where the registers allow): the fragment reads are software-pipelined IN PLACE over the fully unrolled step --
    // the A triple of row block i+1 is read while block i multiplies (one extra register set), and each B triple of the
    // next tap pair is read into its own registers right after its last use in the current pair.  Only the first reads of a
    // step wait on LDS latency, so a wave keeps the matrix pipe busy on its own while its SIMD partner is staging.  The
    // MFMAs of every accumulator are issued in the same order either way: results are bit-identical.
    // (round 6: the one-term bf16 flavour too -- its plain loop read a fragment, waited, multiplied four times: matrix pipe 0.39-0.41 busy,
    // profiles/r05_pmc_mfma_utilisation_cfg3_bf16.csv)
    constexpr bool PIPE = P::NS <= 2;
    auto mma_step = [&](int q, int abuf, int xbuf) {
        const unsigned char* A = Abuf + abuf * ASTEP + a_lane;
        const unsigned char* X = Xbase + xbuf * G::X_BYTES + x_lane;
        auto read_b = [&](int pp, int j, vec8 (&f)[P::NS]) {
            const int row = G::rowmap((wn * WCOLS + j * 16 + l15) * S + q * TP + 2 * pp + (kq >> 1));
#pragma unroll
            for (int k = 0; k < P::NS; ++k) f[k] = *reinterpret_cast<const vec8*>(X + (k * 2 * G::RP + row) * 16);
        };
        auto read_a = [&](int pp, int i, vec8 (&f)[P::NS]) {
#pragma unroll
            for (int k = 0; k < P::NS; ++k) f[k] = *reinterpret_cast<const vec8*>(A + ((k * TP + 2 * pp) * 2 * PB_M + i * 16) * 16);
        };
        auto block = [&](int i, int j, const vec8 (&af)[P::NS], const vec8 (&bf)[P::NS]) {
            floatx4 c = acc[i][j];                  // ONE accumulator, smallest terms first (a dependent chain of the same MFMA
            if constexpr (P::NS == 3) {             // issues back to back on gfx950: nothing to interleave for)
                c = P::mfma(af[2], bf[0], c);       // lo * hi
                c = P::mfma(af[0], bf[2], c);       // hi * lo
                c = P::mfma(af[1], bf[1], c);       // mid * mid
            }
            if constexpr (P::NS > 1) {
                c = P::mfma(af[1], bf[0], c);       // mid * hi   (fp16: lo * hi)
                c = P::mfma(af[0], bf[1], c);       // hi * mid   (fp16: hi * lo)
            }
            acc[i][j] = P::mfma(af[0], bf[0], c);   // hi * hi
        };
        if constexpr (PIPE) {
            vec8 bfr[NJ][P::NS], af[2][P::NS];
#pragma unroll
            for (int j = 0; j < NJ; ++j) read_b(0, j, bfr[j]);
            read_a(0, 0, af[0]);
#pragma unroll
            for (int pp = 0; pp < TP / 2; ++pp) {
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    const int cur = (pp * MI + i) & 1;
                    const bool more_a = i < MI - 1 || pp + 1 < TP / 2;
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        __builtin_amdgcn_sched_barrier(0);
                        block(i, j, af[cur], bfr[j]);
                        __builtin_amdgcn_sched_barrier(0);
                        // hipcc waits for ALL outstanding LDS reads before the first MFMA of a row block (lgkmcnt(0), not a
                        // counted wait), so the next reads are issued BEHIND that first block: they then have 9 MFMAs to land
                        if (j == 0 && more_a) read_a(i < MI - 1 ? pp : pp + 1, (i + 1) % MI, af[cur ^ 1]);
                        if (i == MI - 1 && pp + 1 < TP / 2) read_b(pp + 1, j, bfr[j]);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        } else {
#pragma unroll 1
            for (int pp = 0; pp < TP / 2; ++pp) {
                vec8 bfr[NJ][P::NS];
#pragma unroll
                for (int j = 0; j < NJ; ++j) read_b(pp, j, bfr[j]);
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    vec8 af[P::NS];
                    read_a(pp, i, af);
#pragma unroll
                    for (int j = 0; j < NJ; ++j) block(i, j, af, bfr[j]);
                }
            }
        }
    };

    // ---- main loop: groups of 16 channels, QS K-steps of TP taps each; ONE barrier per K-step ------------------
    //   step (g, q):  [q >= 1: stagers split chunk q-1 of group g+1 and write it to X[(g+1) & 1] -- nobody reads that buffer]
    //                 [q < QS-1: stagers issue the global loads of chunk q of group g+1]   DMA of the next step's weights
    //                 MFMAs on Abuf[step & 1], X[g & 1]
    //                 wait for the DMA (and the chunk loads), barrier
    const int ng = a.n_groups;
    dma_weights(0, 0);
    if constexpr (XIMG) {
#pragma unroll 1
        for (int q = 0; q < QS; ++q) dma_x(0, 0, q);
    } else {
#pragma unroll 1
        for (int c = 0; c < G::NCHUNK; ++c) { load_x(0, c); commit_x(0, c); }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // The SIMD's matrix pipe goes to the older wave whenever both are ready, so without help the younger wave (which multiplies
    // FIRST and stages LAST) finishes its MFMAs late and its staging then runs with the pipe idle.  Static priority for the
    // younger half reverses that: it multiplies at full speed, stages beside the older wave's MFMAs, and the older wave's
    // MFMAs fill the gaps (measured with s_memtime stamps: 5 900 -> 4 100 cycles per K-step at stride 1).
    if (!older) __builtin_amdgcn_s_setprio(1);
    int step = 0;
    for (int g = 0; g < ng; ++g) {
        const bool more = g + 1 < ng;
#pragma unroll 1
        for (int q = 0; q < QS; ++q, ++step) {
            const bool prefetch = q + 1 < QS || more;       // there is a next K-step: fetch its weights
            const bool chunk = more && q < QS - 1;          // chunk q of the next group's input tile is staged in this step
            if (older) {
                // retire the chunk loaded during the previous step, issue this step's loads and the DMA, THEN the MFMAs:
                // the SIMD partner (a younger wave) has the matrix pipe to itself meanwhile
                if constexpr (XIMG) {
                    if (more) dma_x(g + 1, (g + 1) & 1, q);
                } else {
                    if (more && q >= 1) commit_x((g + 1) & 1, q - 1);
                    if (chunk) load_x(g + 1, q);
                }
                if (prefetch) dma_weights(step + 1, (step + 1) & 1);
            } else if (!XIMG && chunk) {
                load_x(g + 1, q);                           // in flight under this wave's own MFMAs
            }
            if (wave_active) mma_step(q, step & 1, g & 1);
            if (!older) {
                // younger waves stage AFTER their MFMAs, beside the older partner's MFMAs
                if constexpr (XIMG) {
                    if (more) dma_x(g + 1, (g + 1) & 1, q);
                } else {
                    if (chunk) commit_x((g + 1) & 1, q);
                }
                if (prefetch) dma_weights(step + 1, (step + 1) & 1);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // next step's weight DMA has landed (this wave's part)
            __syncthreads();               // ... every wave's part, the new input tile is written, this step's reads are done
        }
        flush_group();                     // the group's block sum joins the total: one rounding at full magnitude per 128 k
    }

    if constexpr (P::NS == 1) {
        // ---- bf16 output: bias + ReLU + min(20), ONE rounding, then out through LDS so that a store instruction writes whole
        // 128-byte row segments (64 frames) instead of 32-byte ones.  Each wave owns WROWS x 64 bf16 = WROWS x 128 bytes of
        // the staging buffers, which nobody reads any more (the last K-step ended with a barrier).
        // Passes of two 16-row MFMA tiles per wave: 32 rows x WCOLS bf16 (row stride + 16 bytes: the four row groups of a fragment land on
        // different banks), out as 16-byte pieces -- 8 (64 frames) or 16 (128 frames) lanes per row.
        constexpr int TSH = WCOLS + 8;                      // staged row stride in bf16 elements
        constexpr int LPR = WCOLS / 8, RPI = 64 / LPR;      // lanes per staged row; rows per copy-out instruction
        unsigned short* const T = reinterpret_cast<unsigned short*>(smem) + wave * (32 * TSH);
        static_assert(8 * 32 * TSH * 2 <= G::LDS_BYTES, "output staging must fit the operand buffers");
        if (wave_active) {
            bf16_t* const yb = reinterpret_cast<bf16_t*>(a.y);
            const int n = n0 + wn * WCOLS + (lane % LPR) * 8;
#pragma unroll
            for (int i0 = 0; i0 < MI; i0 += 2) {
#pragma unroll
                for (int ii = 0; ii < 2; ++ii) {
                    const int i = i0 + ii;
                    if (i >= MI) break;
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        const bool live = n0 + wn * WCOLS + j * 16 + l15 < a.frames_out;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int m = m0 + wm * WROWS + i * 16 + kq * 4 + r;
                            const float v = (live && m < a.c_out) ? relu_clamp(acc[i][j][r] + a.bias[m]) : 0.f;
                            T[(ii * 16 + kq * 4 + r) * TSH + j * 16 + l15] = __builtin_bit_cast(unsigned short, static_cast<__bf16>(v));
                        }
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's LDS writes are done (wave-private tile); also keeps
                                                                     // the compiler from moving the differently-typed reads above them
                const int rows_pass = (MI - i0 >= 2) ? 32 : 16;
#pragma unroll
                for (int it = 0; it < 32 / RPI; ++it) {
                    const int ml = it * RPI + lane / LPR, m = m0 + wm * WROWS + i0 * 16 + ml;
                    if (ml < rows_pass && m < a.c_out && n < a.ld_out) {
                        const u4v t = *reinterpret_cast<const u4v*>(T + ml * TSH + (lane % LPR) * 8);
                        *reinterpret_cast<u4v*>(yb + (static_cast<size_t>(b) * a.c_out + m) * a.ld_out + n) = t;
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the reads are done before the next pass overwrites the tile
            }
        }
        return;
    } else {
    // ---- epilogue without skips (every downsample conv of the model): bias + ReLU + min(20), then out through LDS so that a store
    // instruction writes four 256-byte row segments (16 lanes x 16 B) instead of four 64-byte ones.  Two 16-row MFMA tiles at a time
    // per wave: 32 rows x 64 frames fp32 (row stride 68 floats: the four row groups of a fragment land 2-way instead of 4-way on the
    // banks) = 8.5 KiB of the staging buffers, which nobody reads any more (the last K-step ended with a barrier).
    if (!a.s0 && !a.s1 && !a.s2 && a.staged_epilogue) {
        constexpr int TS = WCOLS + 4;                       // (64 + 4: the four row groups of a fragment land 2-way instead of 4-way on the banks)
        constexpr int FQ = WCOLS / 4, RI = 64 / FQ, NIT = 32 / RI;   // frame quads per staged row; rows per copy-out instruction; instructions per pass
        float* const T = reinterpret_cast<float*>(smem) + wave * (32 * TS);
        static_assert(8 * 32 * TS * 4 <= G::LDS_BYTES, "output staging must fit the operand buffers");
        if (!wave_active) return;
        const int nc = n0 + wn * WCOLS + (lane % FQ) * 4;
        // Statistics by-product (round 5; a.part != NULL): per frame the (mean, M2) over each 16-ROW UNIT of the output -- one MFMA row
        // tile, 16 channels aligned to 16 -- written to part[unit][batch][2][ld_out]; stats_finalize_kernel merges the units in ascending
        // order.  The unit, not the workgroup's tile, is the granule: its sums are the same arithmetic whatever tile (64 ... 160 rows, 128 or
        // 256 frames, chosen per batch size) the launch runs with, so the statistics -- like y -- do not depend on the batch an utterance
        // sits in.  A lane takes ONE frame of the staged tile and walks down its column (ds_read_b32, consecutive lanes on consecutive
        // banks): shifted sums S1 = sum (v - c), S2 = sum (v - c)^2 with c = the unit's first row, so nothing cancels; no cross-lane step.
        // (128-frame tiles: the two halves of the wave take the pass's two units side by side.)
        const bool want_stats = a.part != nullptr;          // (kernel-uniform)
        const int scol = lane % WCOLS, usel = lane / WCOLS; // this lane's frame (and first unit) in the statistics pass
        const int ncol = n0 + wn * WCOLS + scol;
#pragma unroll
        for (int i0 = 0; i0 < MI; i0 += 2) {
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
                const int i = i0 + ii;
                if (i >= MI) break;
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const bool live = n0 + wn * WCOLS + j * 16 + l15 < a.frames_out;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int m = m0 + wm * WROWS + i * 16 + kq * 4 + r;
                        float v = 0.f;
                        if (live && m < a.c_out) {
                            float sum = tot[i][j][r];
                            if constexpr (P::SCALED) sum = sum * x_inv * a.w_inv_scale[m];
                            v = relu_clamp(sum + a.bias[m]);
                        }
                        T[(ii * 16 + kq * 4 + r) * TS + j * 16 + l15] = v;
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // wave-private tile: this wave's writes are done
            const int rows_pass = (MI - i0 >= 2) ? 32 : 16;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int ml = it * RI + lane / FQ;
                const int m = m0 + wm * WROWS + i0 * 16 + ml;
                if (ml < rows_pass && m < a.c_out && nc < a.ld_out) {
                    const floatx4 t = *reinterpret_cast<const floatx4*>(T + ml * TS + (lane % FQ) * 4);
                    *reinterpret_cast<floatx4*>(a.y + (static_cast<size_t>(b) * a.c_out + m) * a.ld_out + nc) = t;
                }
            }
            if (want_stats) {
#pragma unroll
                for (int u0 = 0; u0 < 2; u0 += 64 / WCOLS) {
                    const int u = u0 + usel;
                    const int mu = m0 + wm * WROWS + (i0 + u) * 16;          // the unit's first channel (a multiple of 16)
                    if (i0 + u < MI && mu < a.c_out) {
                        const int nrows = min(16, a.c_out - mu);
                        const float c = T[(u * 16) * TS + scol];
                        float s1 = 0.f, s2 = 0.f;
#pragma unroll
                        for (int r = 1; r < 16; ++r) {
                            const float d = (r < nrows) ? T[(u * 16 + r) * TS + scol] - c : 0.f;
                            s1 += d;
                            s2 = __builtin_fmaf(d, d, s2);
                        }
                        if (ncol < a.ld_out) {
                            const float inv = 1.0f / static_cast<float>(nrows);
                            float* prow = a.part + (static_cast<size_t>(mu >> 4) * a.batch + b) * 2 * a.ld_out + ncol;
                            prow[0] = c + s1 * inv;
                            prow[a.ld_out] = s2 - s1 * s1 * inv;
                        }
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the reads are done before the next pass overwrites the tile
        }
        return;
    }
    // ---- epilogue with skips: bias + ReLU + min(20) + skips; a store covers 4 rows x 16 consecutive frames -----------------
    if (!wave_active) return;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int n = n0 + wn * WCOLS + j * 16 + l15;
            if (n >= a.ld_out) continue;
            const bool live = n < a.frames_out;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * WROWS + i * 16 + kq * 4 + r;
                if (m >= a.c_out) continue;
                float sum = tot[i][j][r];
                if constexpr (P::SCALED) sum = sum * x_inv * a.w_inv_scale[m];  // exact: powers of two, applied one after the other
                                                                               // (their product alone could leave fp32's range)
                float v = relu_clamp(sum + a.bias[m]);
                const size_t off = (static_cast<size_t>(b) * a.c_out + m) * a.ld_out + n;
                if (a.s0) v += a.s0[off];
                if (a.s1) v += a.s1[off];
                if (a.s2) v += a.s2[off];
                a.y[off] = live ? v : 0.f;
            }
        }
    }
    }
}

// Tile order (round 4): frame-tile major.  Round 1-3's row-tile-major order kept a row tile's weights in an XCD's L2 and streamed the
// whole operand image past them once per row tile: 3.0-5.3 x the algorithmic HBM bytes.  Frame-tile major, the
// 5-19 row tiles of one frame tile run next to each other on one XCD and share its image through L2 -- one pass over the image -- while
// the weights (15-38 MB, every XCD wants all of them all the time) come from the last-level cache.  Same tiles, same sums; same-box
// A/B at 64 x 1000: 9 434 / 9 501 against 9 328 / 9 411 utterances/s (+1 %), conv 3 692 -> 671 us.

template <class P, int S>
static int launch_packed(PackedConvArgs a, hipStream_t stream)
{
    using G = GeoP<P, S>;
    constexpr int PB_M = G::PBM;
    constexpr bool HAS_LNX = !P::SCALED;      // the scaled scheme needs the range of the normalised tensor: no LayerNorm on load
    static const hipError_t attr0 = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_conv_split_kernel<P, S, false>),
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
    static const hipError_t attr1 = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_conv_split_kernel<P, S, HAS_LNX>),
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
    const hipError_t attr = attr0 != hipSuccess ? attr0 : attr1;
    if (attr != hipSuccess) {
        set_error("%s: cannot reserve %d bytes of LDS: %s", P::NAME, G::LDS_BYTES, hipGetErrorString(attr));
        return static_cast<int>(attr);
    }
    NBASR_REQUIRE(HAS_LNX || !a.ln_x.stats, NBASR_EINVAL, "%s: this scheme takes no deferred LayerNorm", P::NAME);
    a.n_mt = (a.c_out + PB_M - 1) / PB_M;
    a.n_nt = (a.ld_out + G::PBN - 1) / G::PBN;
    const long long nwg = static_cast<long long>(a.n_mt) * a.n_nt * a.batch;
    NBASR_REQUIRE(nwg < (1ll << 31), NBASR_EINVAL, "%s: too many tiles (%lld)", P::NAME, nwg);
    if (a.ln_x.stats)
        hipLaunchKernelGGL((gemm_conv_split_kernel<P, S, HAS_LNX>), dim3(static_cast<unsigned>(nwg)), dim3(PB_THREADS), G::LDS_BYTES, stream, a);
    else
        hipLaunchKernelGGL((gemm_conv_split_kernel<P, S, false>), dim3(static_cast<unsigned>(nwg)), dim3(PB_THREADS), G::LDS_BYTES, stream, a);
    return launch_status(P::NAME);
}

static inline int rows_to_mi(int row_tile) { return row_tile == 128 ? 4 : (row_tile == 160 ? 5 : (row_tile == 96 ? 3 : (row_tile == 64 ? 2 : 0))); }

// image-path kernel of scheme P at a given row tile (the only kernel the plain-bf16 scheme has)
template <class P, int S, int MI, int NJ = 4>
static int launch_image(PackedConvArgs a, hipStream_t stream)
{
    using G = GeoP<P, S, MI, NJ>;
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_conv_split_kernel<P, S, false, true, MI, NJ>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
    if (attr != hipSuccess) {
        set_error("%s: cannot reserve %d bytes of LDS: %s", P::NAME, G::LDS_BYTES, hipGetErrorString(attr));
        return static_cast<int>(attr);
    }
    a.n_mt = (a.c_out + G::PBM - 1) / G::PBM;
    a.n_nt = (a.ld_out + G::PBN - 1) / G::PBN;
    const long long nwg = static_cast<long long>(a.n_mt) * a.n_nt * a.batch;
    NBASR_REQUIRE(nwg < (1ll << 31), NBASR_EINVAL, "%s: too many tiles (%lld)", P::NAME, nwg);
    hipLaunchKernelGGL((gemm_conv_split_kernel<P, S, false, true, MI, NJ>), dim3(static_cast<unsigned>(nwg)), dim3(PB_THREADS), G::LDS_BYTES, stream, a);
    return launch_status(P::NAME);
}

template <class P>
static size_t packed_bytes(int c_out, int c_in, int kernel, int row_tile = 128)
{
    const int mi = rows_to_mi(row_tile);
    if (c_out <= 0 || c_in <= 0 || kernel != PB_TAPS || mi == 0 || (mi != 4 && !has_image_path<P>())) return 0;
    const size_t PB_M = pb_rows(mi), n_mt = (c_out + PB_M - 1) / PB_M, n_groups = (c_in + PB_CI - 1) / PB_CI;
    return n_mt * n_groups * pb_group_bytes<P>(mi) + (P::SCALED ? 2 * n_mt * PB_M * sizeof(float) : 0);   // + row scales and inverses
}

template <class P>
static int pack_impl(const float* w, void* packed, int c_out, int c_in, int kernel, int stride, nbasr_stream_t stream, int row_tile = 128)
{
    clear_error();
    const int mi = rows_to_mi(row_tile);
    NBASR_REQUIRE(mi != 0 && (mi == 4 || has_image_path<P>()), NBASR_EINVAL, "nbasr_pack_dense_weights: row_tile=%d unsupported (128, or 64 / 96 / 160 for the image-path schemes)", row_tile);
    const int PB_M = pb_rows(mi);
    NBASR_REQUIRE(c_out > 0 && c_in > 0, NBASR_EINVAL, "nbasr_pack_dense_weights: bad sizes");
    NBASR_REQUIRE(kernel == PB_TAPS && (stride == 1 || stride == 2), NBASR_EINVAL,
                  "nbasr_pack_dense_weights: (kernel=%d, stride=%d) unsupported (the downsample convs have k=8, s in {1,2})", kernel, stride);
    NBASR_REQUIRE(w && packed, NBASR_ENULL, "nbasr_pack_dense_weights: NULL pointer");
    NBASR_REQUIRE(aligned16(packed), NBASR_EALIGN, "nbasr_pack_dense_weights: packed buffer must be 16-byte aligned");
    const int n_mt = (c_out + PB_M - 1) / PB_M, n_groups = (c_in + PB_CI - 1) / PB_CI;
    float* scales = nullptr;
    if (P::SCALED) {
        scales = reinterpret_cast<float*>(static_cast<unsigned char*>(packed) + static_cast<size_t>(n_mt) * n_groups * pb_group_bytes<P>(mi));
        hipLaunchKernelGGL(weight_row_scales_kernel, dim3(n_mt * PB_M), dim3(256), 0, as_stream(stream), w, scales, c_out,
                           c_in * PB_TAPS, n_mt * PB_M);
    }
    hipLaunchKernelGGL(pack_dense_weights_kernel<P>, dim3(2048), dim3(256), 0, as_stream(stream), w,
                       static_cast<unsigned short*>(packed), scales, c_out, c_in, n_mt, n_groups, P::taps_per_step(stride), PB_M);
    return launch_status("nbasr_pack_dense_weights");
}

template <class P>
static int dense_packed_impl(const float* x, const void* packed_w, const float* bias, const float* skip0,
                             const float* skip1, const float* skip2, float* y, int batch, int c_in,
                             int frames_in, int ld_in, int c_out, int ld_out, int kernel, int stride,
                             const nbasr_deferred_ln* ln, const float* x_absmax, nbasr_stream_t stream, bool x_is_image = false,
                             int row_tile = 128, const float* x_range = nullptr, int sel_want = -1, float* stats_part = nullptr,
                             int frame_tile = 256)
{
    clear_error();
    const int mi = rows_to_mi(row_tile);
    NBASR_REQUIRE(mi == 4 || ((mi == 5 || mi == 3 || mi == 2) && x_is_image && P::SCALED), NBASR_EINVAL,
                  "%s: row_tile=%d unsupported (128; 64, 96 and 160 for the fp16 image path)", P::NAME, row_tile);
    const int PB_M = pb_rows(mi);
    NBASR_REQUIRE(frame_tile == 256 || (frame_tile == 128 && x_is_image && P::SCALED), NBASR_EINVAL,
                  "%s: frame_tile=%d unsupported (256; 128 for the fp16 image path)", P::NAME, frame_tile);
    NBASR_REQUIRE(batch >= 0 && c_in > 0 && c_out > 0 && frames_in >= 0, NBASR_EINVAL, "%s: bad sizes", P::NAME);
    NBASR_REQUIRE(kernel == PB_TAPS && (stride == 1 || stride == 2), NBASR_EINVAL,
                  "%s: (kernel=%d, stride=%d) unsupported; packed path covers k=8, s in {1,2}", P::NAME, kernel, stride);
    const int frames_out = (frames_in + stride - 1) / stride;
    NBASR_REQUIRE(ld_in >= frames_in, NBASR_EINVAL, "%s: ld_in=%d < frames_in=%d", P::NAME, ld_in, frames_in);
    NBASR_REQUIRE(ld_in % 4 == 0, NBASR_EALIGN, "%s: ld_in=%d must be a multiple of 4", P::NAME, ld_in);
    NBASR_REQUIRE(ld_out >= frames_out && ld_out % 4 == 0, NBASR_EALIGN,
                  "%s: ld_out=%d must be >= %d output frames and a multiple of 4", P::NAME, ld_out, frames_out);
    if (batch == 0 || frames_out == 0) return NBASR_OK;
    NBASR_REQUIRE(x && packed_w && bias && y, NBASR_ENULL, "%s: x, packed_w, bias, y must be non-NULL", P::NAME);
    NBASR_REQUIRE(aligned16(packed_w) && aligned16(x), NBASR_EALIGN, "%s: x and packed weights must be 16-byte aligned", P::NAME);
    PackedConvArgs a{};
    a.x = x; a.wp = static_cast<const unsigned char*>(packed_w); a.bias = bias; a.s0 = skip0; a.s1 = skip1; a.s2 = skip2; a.y = y;
    a.c_in = c_in; a.frames_in = frames_in; a.ld_in = ld_in; a.c_out = c_out; a.frames_out = frames_out; a.ld_out = ld_out;
    a.lpad = pad_left(kernel, 1, stride); a.n_groups = (c_in + PB_CI - 1) / PB_CI; a.batch = batch;
    NBASR_REQUIRE(!ln || (ln->stats && ln->gamma && ln->beta), NBASR_ENULL, "%s: deferred LayerNorm needs stats, gamma and beta", P::NAME);
    a.ln_x = ln_ref(ln, true);
    a.x_range = x_range;
    a.sel_want = x_range ? sel_want : -1;
    a.staged_epilogue = 1;
    NBASR_REQUIRE(!stats_part || (!skip0 && !skip1 && !skip2 && aligned16(stats_part)), NBASR_EINVAL,
                  "%s: the statistics by-product exists for the skip-free convolution only (stats_part 16-byte aligned)", P::NAME);
    a.part = stats_part;
    if (P::SCALED) {
        NBASR_REQUIRE(x_absmax || x_range, NBASR_ENULL, "%s: x_absmax (per-utterance bound of |x|) must be non-NULL", P::NAME);
        a.x_absmax = x_absmax;
        a.x_is_image = x_is_image ? 1 : 0;
        a.w_inv_scale = reinterpret_cast<const float*>(a.wp + static_cast<size_t>((c_out + PB_M - 1) / PB_M) * a.n_groups * pb_group_bytes<P>(mi))
                        + static_cast<size_t>((c_out + PB_M - 1) / PB_M) * PB_M;
    }
    if constexpr (has_image_path<P>() && P::SCALED) {
        if (x_is_image) {
            // 160-row tiles where 128 rows leave a mostly empty last row tile or a partial last round of workgroups; 64- / 96-row tiles:
            // more workgroups where a small batch leaves CUs without one (the executor's round count decides)
            hipStream_t s = as_stream(stream);
            if (frame_tile == 128) {
                // 128-frame tiles (round 5): twice the workgroups where a small batch leaves compute units without one
                switch (mi) {
                    case 5: return stride == 1 ? launch_image<P, 1, 5, 2>(a, s) : launch_image<P, 2, 5, 2>(a, s);
                    case 3: return stride == 1 ? launch_image<P, 1, 3, 2>(a, s) : launch_image<P, 2, 3, 2>(a, s);
                    case 2: return stride == 1 ? launch_image<P, 1, 2, 2>(a, s) : launch_image<P, 2, 2, 2>(a, s);
                    default: return stride == 1 ? launch_image<P, 1, 4, 2>(a, s) : launch_image<P, 2, 4, 2>(a, s);
                }
            }
            switch (mi) {
                case 5: return stride == 1 ? launch_image<P, 1, 5>(a, s) : launch_image<P, 2, 5>(a, s);
                case 3: return stride == 1 ? launch_image<P, 1, 3>(a, s) : launch_image<P, 2, 3>(a, s);
                case 2: return stride == 1 ? launch_image<P, 1, 2>(a, s) : launch_image<P, 2, 2>(a, s);
                default: return stride == 1 ? launch_image<P, 1, 4>(a, s) : launch_image<P, 2, 4>(a, s);
            }
        }
    }
    return stride == 1 ? launch_packed<P, 1>(a, as_stream(stream)) : launch_packed<P, 2>(a, as_stream(stream));
}

}  // namespace nbasr

using namespace nbasr;

extern "C" size_t nbasr_split_image_bytes(int batch, int channels, int ld)
{
    if (batch <= 0 || channels <= 0 || ld < 0) return 0;
    return static_cast<size_t>(batch) * ((channels + PB_CI - 1) / PB_CI) * 4 * (static_cast<size_t>(ld) + 1) * 16;
}

extern "C" size_t nbasr_packed_dense_weights_bytes(int scheme, int c_out, int c_in, int kernel, int row_tile)
{
    switch (scheme) {
        case NBASR_DENSE_BF16X3: return packed_bytes<SplitBf16x3>(c_out, c_in, kernel, row_tile);
        case NBASR_DENSE_F16X2:  return packed_bytes<SplitF16x2>(c_out, c_in, kernel, row_tile);
        case NBASR_DENSE_BF16:   return packed_bytes<PlainBf16>(c_out, c_in, kernel, row_tile);
        default: return 0;
    }
}

extern "C" int nbasr_pack_dense_weights(int scheme, const float* w, void* packed, int c_out, int c_in, int kernel, int stride, int row_tile,
                                        nbasr_stream_t stream)
{
    switch (scheme) {
        case NBASR_DENSE_BF16X3: return pack_impl<SplitBf16x3>(w, packed, c_out, c_in, kernel, stride, stream, row_tile);
        case NBASR_DENSE_F16X2:  return pack_impl<SplitF16x2>(w, packed, c_out, c_in, kernel, stride, stream, row_tile);
        case NBASR_DENSE_BF16:   return pack_impl<PlainBf16>(w, packed, c_out, c_in, kernel, stride, stream, row_tile);
        default: break;
    }
    clear_error();
    NBASR_REQUIRE(false, NBASR_EINVAL, "nbasr_pack_dense_weights: unknown scheme %d", scheme);
    return NBASR_EINVAL;
}

static int dense_bf16_image_impl(const void* x_image, const void* packed_w, const float* bias, void* y, int batch, int c_in,
                                 int frames_in, int ld_in, int c_out, int ld_out, int kernel, int stride, int row_tile, int frame_tile,
                                 nbasr_stream_t stream)
{
    using P = PlainBf16;
    clear_error();
    const int mi = rows_to_mi(row_tile);
    NBASR_REQUIRE(mi == 4 || mi == 5, NBASR_EINVAL, "%s: row_tile=%d unsupported (128 or 160)", P::NAME, row_tile);
    NBASR_REQUIRE(batch >= 0 && c_in > 0 && c_out > 0 && frames_in >= 0, NBASR_EINVAL, "%s: bad sizes", P::NAME);
    NBASR_REQUIRE(kernel == PB_TAPS && (stride == 1 || stride == 2), NBASR_EINVAL,
                  "%s: (kernel=%d, stride=%d) unsupported; the downsample convs have k=8, s in {1,2}", P::NAME, kernel, stride);
    const int frames_out = (frames_in + stride - 1) / stride;
    NBASR_REQUIRE(ld_in >= frames_in, NBASR_EINVAL, "%s: ld_in=%d < frames_in=%d", P::NAME, ld_in, frames_in);
    NBASR_REQUIRE(ld_out >= frames_out && ld_out % 8 == 0, NBASR_EALIGN,
                  "%s: ld_out=%d must be >= %d output frames and a multiple of 8", P::NAME, ld_out, frames_out);
    if (batch == 0 || frames_out == 0) return NBASR_OK;
    NBASR_REQUIRE(x_image && packed_w && bias && y, NBASR_ENULL, "%s: x_image, packed_w, bias, y must be non-NULL", P::NAME);
    NBASR_REQUIRE(aligned16(packed_w) && aligned16(x_image) && aligned16(y), NBASR_EALIGN, "%s: pointers must be 16-byte aligned", P::NAME);
    PackedConvArgs a{};
    a.x = static_cast<const float*>(x_image); a.wp = static_cast<const unsigned char*>(packed_w); a.bias = bias;
    a.y = static_cast<float*>(y);
    a.c_in = c_in; a.frames_in = frames_in; a.ld_in = ld_in; a.c_out = c_out; a.frames_out = frames_out; a.ld_out = ld_out;
    a.lpad = pad_left(kernel, 1, stride); a.n_groups = (c_in + PB_CI - 1) / PB_CI; a.batch = batch;
    a.ln_x = LnRef{nullptr, nullptr, nullptr};
    a.x_is_image = 1;
    hipStream_t s = as_stream(stream);
    if (frame_tile == 512) {
        // 512-frame tiles (round 6): a wave's register tile is 16 MI rows x 128 frames -- a B fragment is re-used MI times as before, an A
        // fragment 8 times instead of 4: 0.33 instead of 0.45 KiB of LDS reads per MFMA, where the LDS read rate bounds this flavour
        if (mi == 5) return stride == 1 ? launch_image<P, 1, 5, 8>(a, s) : launch_image<P, 2, 5, 8>(a, s);
        return stride == 1 ? launch_image<P, 1, 4, 8>(a, s) : launch_image<P, 2, 4, 8>(a, s);
    }
    if (mi == 5) return stride == 1 ? launch_image<P, 1, 5>(a, s) : launch_image<P, 2, 5>(a, s);
    return stride == 1 ? launch_image<P, 1, 4>(a, s) : launch_image<P, 2, 4>(a, s);
}

// ONE entry point for the packed k = 8 convolution (nbasr.h: the schemes, the two operand forms and the per-utterance routing)
extern "C" int nbasr_dense_conv1d_packed(int scheme, const void* x, int x_is_image, const float* x_absmax, const float* x_range,
                                         const void* packed_w, const float* bias, const float* skip0, const float* skip1,
                                         const float* skip2, void* y, int batch, int c_in, int frames_in, int ld_in, int c_out,
                                         int ld_out, int kernel, int stride, int row_tile, int frame_tile, const nbasr_deferred_ln* ln,
                                         float* stats_part, nbasr_stream_t stream)
{
    clear_error();
    const float* xf = static_cast<const float*>(x);
    float* yf = static_cast<float*>(y);
    const bool image = x_is_image != 0;
    switch (scheme) {
        case NBASR_DENSE_BF16X3:
            NBASR_REQUIRE(!image && !x_absmax, NBASR_EINVAL, "nbasr_dense_conv1d_packed: the bf16x3 scheme reads a plain fp32 tensor and takes no bound");
            // with x_range: only the EXTREME utterances (the fp16 leg computes the others on the same output)
            return dense_packed_impl<SplitBf16x3>(xf, packed_w, bias, skip0, skip1, skip2, yf, batch, c_in, frames_in, ld_in, c_out, ld_out,
                                                  kernel, stride, ln, nullptr, stream, false, row_tile, x_range, 1, stats_part, frame_tile);
        case NBASR_DENSE_F16X2:
            NBASR_REQUIRE(!ln, NBASR_EINVAL, "nbasr_dense_conv1d_packed: the fp16x2 scheme takes no pending LayerNorm (nbasr_layernorm_split_image writes its operand)");
            NBASR_REQUIRE(!(image && (skip0 || skip1 || skip2)), NBASR_EINVAL, "nbasr_dense_conv1d_packed: the image path takes no skips");
            NBASR_REQUIRE(!(x_absmax && x_range), NBASR_EINVAL, "nbasr_dense_conv1d_packed: give x_absmax or x_range, not both");
            return dense_packed_impl<SplitF16x2>(xf, packed_w, bias, skip0, skip1, skip2, yf, batch, c_in, frames_in, ld_in, c_out, ld_out,
                                                 kernel, stride, nullptr, x_absmax, stream, image, row_tile, x_range, 0, stats_part, frame_tile);
        case NBASR_DENSE_BF16:
            NBASR_REQUIRE(frame_tile == 256 || frame_tile == 512, NBASR_EINVAL, "nbasr_dense_conv1d_packed: the bf16 scheme has 256- and 512-frame tiles");
            NBASR_REQUIRE(image && !x_absmax && !x_range && !ln && !skip0 && !skip1 && !skip2 && !stats_part, NBASR_EINVAL,
                          "nbasr_dense_conv1d_packed: the bf16 scheme reads nbasr_bf16_image's operand image only (no bound, range, LayerNorm, skips or statistics)");
            return dense_bf16_image_impl(x, packed_w, bias, y, batch, c_in, frames_in, ld_in, c_out, ld_out, kernel, stride, row_tile, frame_tile, stream);
        default: break;
    }
    NBASR_REQUIRE(false, NBASR_EINVAL, "nbasr_dense_conv1d_packed: unknown scheme %d", scheme);
    return NBASR_EINVAL;
}
