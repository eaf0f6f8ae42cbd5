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
// GEMM view per utterance: M = c_out, N = output frames, K = (c_in, tap).  The 32 k of one MFMA are 16 input CHANNELS of
// TWO consecutive taps (a lane holds 8 consecutive k = 8 channels of one tap):
//     D[co][t] += sum_{ci<16} W[co][g*16+ci][tap] * x[g*16+ci][t*stride + tap - lpad]      for every (group g, tap)
//  * weights are split and re-laid-out ONCE (nbasr_pack_dense_weights) into the exact LDS image of each
//    (row tile, channel group, tap quad): [split][tap][ci half][128 rows][8 ci] bf16 = 48 KiB, so a K-step's weights are a
//    straight 48 KiB copy done by LDS-DMA (global_load_lds_dwordx4, no VGPRs), double-buffered;
//  * the input tile of a channel group is fetched as aligned 4-frame quads, split on the fly and stored TRANSPOSED
//    [ci half][frame][8 ci] so a B fragment (8 consecutive channels of one frame) is one aligned, bank-conflict-free
//    ds_read_b128; it is staged once per group and reused by all 8 taps (sliding window resolved by the row index;
//    stride-2 rows are de-interleaved by parity so the 16 lanes of a fragment read hit consecutive rows);
//  * one 512-thread workgroup per CU: 128 x 256 tile / 8 waves (2 x 4, 64 x 64 each = 4 x 4 MFMA tiles), K-step =
//    16 channels x 4 taps (2 for stride 2), one barrier per step; the two halves of the workgroup run a step in opposite
//    order (stage-then-multiply / multiply-then-stage) so a wave's memory phase sits beside its SIMD partner's MFMAs;
//  * the hi*hi products accumulate in their own register set, so the large running sum is rounded once per 32 k.
#include "common.h"

#include <type_traits>

namespace nbasr {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int PB_M = 128, PB_N = 256, PB_CI = 16, PB_TAPS = 8;   // tile, channels per group, conv taps
constexpr int PB_THREADS = 512;                                              // 8 waves: 2 (rows) x 4 (frames), 64 x 64 each
constexpr int PB_GROUP_BYTES = 3 * PB_TAPS * PB_M * PB_CI * 2;               // 98304: packed weights of one (row tile, channel group)
__host__ __device__ constexpr int pb_taps_per_step(int stride) { return stride == 1 ? 4 : 2; }   // LDS budget: 160 KiB

template <int S>
struct GeoP {
    static constexpr int TP = pb_taps_per_step(S);               // taps per K-step
    static constexpr int QSTEPS = PB_TAPS / TP;                  // K-steps per channel group
    static constexpr int A_STEP_BYTES = 3 * TP * PB_M * PB_CI * 2;
    static constexpr int XR = (PB_N - 1) * S + PB_TAPS;          // input frames needed per channel
    static constexpr int XRH = (XR + 1) / 2;                     // rows per parity plane (stride 2)
    static constexpr int ROWS = (S == 1) ? XR : 2 * XRH;         // rows per split plane
    static constexpr int X_BYTES = 3 * ROWS * PB_CI * 2;
    // the tile is fetched as ALIGNED 4-frame quads (one global_load_dwordx4 per channel): NQUADS covers XR rows at any
    // misalignment of the tile's first frame; an item = (quad, channel pair), 8 consecutive quads x 8 pairs per wave
    static constexpr int NQUADS = ((XR + 3 + 3) / 4 + 7) / 8 * 8;
    static constexpr int XITEMS = NQUADS * (PB_CI / 2);
    static constexpr int NCHUNK = QSTEPS - 1;                    // the next group's tile is staged in QSTEPS-1 chunks
    static constexpr int XI = (XITEMS + PB_THREADS * NCHUNK - 1) / (PB_THREADS * NCHUNK);   // items per thread per chunk
    static constexpr int LDS_BYTES = 2 * A_STEP_BYTES + 2 * X_BYTES;   // weights and input tile both double-buffered
    __device__ static constexpr int rowmap(int row) { return (S == 1) ? row : (row & 1) * XRH + (row >> 1); }
};

struct PackedConvArgs {
    const float* x; const unsigned char* wp; const float* bias;
    const float* s0; const float* s1; const float* s2;
    float* y;
    int c_in, frames_in, ld_in, c_out, frames_out, ld_out, lpad;
    int n_groups, n_mt, n_nt, batch;
    LnRef ln_x;                  // pending LayerNorm of the input (deferred normalisation, nbasr.h)
};

__device__ __forceinline__ void split3(float v, __bf16& hi, __bf16& mid, __bf16& lo) {
    hi = static_cast<__bf16>(v);
    const float r1 = v - static_cast<float>(hi);
    mid = static_cast<__bf16>(r1);
    const float r2 = r1 - static_cast<float>(mid);
    lo = static_cast<__bf16>(r2);
}

__device__ __forceinline__ unsigned pack2(__bf16 a, __bf16 b) {
    return static_cast<unsigned>(__builtin_bit_cast(unsigned short, a)) |
           (static_cast<unsigned>(__builtin_bit_cast(unsigned short, b)) << 16);
}

// ---- one-time weight split + re-layout -------------------------------------------------------------------------
// packed element (mt, g, q, split, tp, half = ci_l/8, co_l, ci_l%8) <- W[mt*128 + co_l][g*16 + ci_l][q*TP + tp]  (zero outside);
// TP = taps per K-step of the kernel that will consume the image (4 for stride 1, 2 for stride 2)
__global__ __launch_bounds__(256) void pack_dense_weights_kernel(const float* __restrict__ w, __bf16* __restrict__ wp,
                                                                 int c_out, int c_in, int n_mt, int n_groups, int PB_TP)
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
        const float v = (co < c_out && ci < c_in) ? w[(static_cast<size_t>(co) * c_in + ci) * PB_TAPS + tap] : 0.f;
        __bf16 s[3];
        split3(v, s[0], s[1], s[2]);
        const size_t step = (static_cast<size_t>(mt) * n_groups + g) * PB_QSTEPS + q;
#pragma unroll
        for (int k = 0; k < 3; ++k)
            wp[((((step * 3 + k) * PB_TP + tp) * 2 + (ci_l >> 3)) * PB_M + co_l) * 8 + (ci_l & 7)] = s[k];
    }
}

// ---- the GEMM -----------------------------------------------------------------------------------------------------
template <int S, bool LNX>
__global__ __launch_bounds__(PB_THREADS, 2) void gemm_conv_bf16x3_kernel(const PackedConvArgs a)
{
    using G = GeoP<S>;
    constexpr int TP = G::TP, QS = G::QSTEPS, ASTEP = G::A_STEP_BYTES;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const Abuf = smem;                       // [2][ASTEP]    weights of the current / next K-step
    unsigned char* const Xbase = smem + 2 * ASTEP;          // [2][X_BYTES]  input tile of the current / next channel group

    // XCD-aware, m-major tile order (as gemm_conv.hip)
    const int nwg = gridDim.x, id = blockIdx.x;
    const int xq = nwg >> 3, xr = nwg & 7, xcd = id & 7;
    const int L = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (id >> 3);
    const int per_m = a.n_nt * a.batch;
    const int mt_i = L / per_m;
    const int rem = L - mt_i * per_m;
    const int b = rem / a.n_nt;
    const int nt_i = rem - b * a.n_nt;
    const int m0 = mt_i * PB_M, n0 = nt_i * PB_N;

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
    const unsigned char* __restrict__ wtile = a.wp + static_cast<size_t>(mt_i) * a.n_groups * PB_GROUP_BYTES;

    // a wave whose whole 64 x 64 tile is out of range issues no MFMAs
    const bool wave_active = (m0 + wm * 64) < a.c_out && (n0 + wn * 64) < a.ld_out;

    // two accumulator sets: `big` only ever receives hi*hi (ONE rounding of the large running sum per 16 k, the
    // accumulation-chain length of a 16-way blocked fp32 sum); the five small cross terms go to `small`
    floatx4 big[4][4], small[4][4];                   // 16 x 16 tiles: row (lane >> 4) * 4 + r, column lane & 15
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) { big[i][j][r] = 0.f; small[i][j][r] = 0.f; }

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
    floatx4 xreg[G::XI][2];
    // item e of a chunk: quad (e >> 6) * 8 + (e & 7) (frames a0 + 4 * quad .. + 3, a0 = tin0 rounded down to a multiple
    // of 4), channel pair (e >> 3) & 7.  Quads are aligned, so each is wholly inside [0, ld_in) or wholly outside; the
    // pitch columns frames_in..ld_in-1 are zero by the layout contract (nbasr.h) and their rstd is 0.
    const int a0 = tin0 & ~3, xoff = tin0 - a0;
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
            unsigned* const col = X + (p >> 2) * G::ROWS * 4 + (p & 3);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = row0 + j;
                if (row < 0 || row >= G::XR) continue;
                __bf16 s0[3], s1[3];
                split3(xreg[i][0][j], s0[0], s0[1], s0[2]);
                split3(xreg[i][1][j], s1[0], s1[1], s1[2]);
                unsigned* dst = col + G::rowmap(row) * 4;
#pragma unroll
                for (int k = 0; k < 3; ++k) dst[k * 2 * G::ROWS * 4] = pack2(s0[k], s1[k]);
            }
        }
    };

    // per-lane fragment bases (bytes).  v_mfma_f32_16x16x32_bf16: lane l supplies row/column l & 15 and k = 8 * (l >> 4) ..
    // + 7; the 32 k of one MFMA are the 16 channels of TWO consecutive taps: k quarter kq = l >> 4 -> tap kq >> 1 of the
    // pair, channel half kq & 1.  Both LDS images are [..][tap][half][row][8 channels] with 16-byte rows, so the 16 lanes
    // of a quarter read 256 contiguous bytes (all 64 banks) and a fragment is one ds_read_b128.
    const int l15 = lane & 15, kq = lane >> 4;
    const int a_lane = (((kq >> 1) * 2 + (kq & 1)) * PB_M + wm * 64 + l15) * 16;
    const int x_lane = (kq & 1) * G::ROWS * 16;

    auto mma_step = [&](int q, int abuf, int xbuf) {
        const unsigned char* A = Abuf + abuf * ASTEP + a_lane;
        const unsigned char* X = Xbase + xbuf * G::X_BYTES + x_lane;
#pragma unroll 1
        for (int pp = 0; pp < TP / 2; ++pp) {
            const int tap = q * TP + 2 * pp + (kq >> 1);
            bf16x8 bfr[4][3];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = G::rowmap((wn * 64 + j * 16 + l15) * S + tap);
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    bfr[j][k] = *reinterpret_cast<const bf16x8*>(X + (k * 2 * G::ROWS + row) * 16);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                bf16x8 af[3];
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    af[k] = *reinterpret_cast<const bf16x8*>(A + ((k * TP + 2 * pp) * 2 * PB_M + i * 16) * 16);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    floatx4 c = small[i][j];
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[2], bfr[j][0], c, 0, 0, 0);   // lo * hi
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], bfr[j][2], c, 0, 0, 0);   // hi * lo
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1], bfr[j][1], c, 0, 0, 0);   // mid * mid
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1], bfr[j][0], c, 0, 0, 0);   // mid * hi
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], bfr[j][1], c, 0, 0, 0);   // hi * mid
                    small[i][j] = c;
                    big[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], bfr[j][0], big[i][j], 0, 0, 0);   // hi * hi
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
#pragma unroll 1
    for (int c = 0; c < G::NCHUNK; ++c) { load_x(0, c); commit_x(0, c); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

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
                if (more && q >= 1) commit_x((g + 1) & 1, q - 1);
                if (chunk) load_x(g + 1, q);
                if (prefetch) dma_weights(step + 1, (step + 1) & 1);
            } else if (chunk) {
                load_x(g + 1, q);                           // in flight under this wave's own MFMAs
            }
            if (wave_active) mma_step(q, step & 1, g & 1);
            if (!older) {
                // younger waves stage AFTER their MFMAs, beside the older partner's MFMAs
                if (chunk) commit_x((g + 1) & 1, q);
                if (prefetch) dma_weights(step + 1, (step + 1) & 1);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // next step's weight DMA has landed (this wave's part)
            __syncthreads();               // ... every wave's part, the new input tile is written, this step's reads are done
        }
    }

    // ---- epilogue: bias + ReLU + min(20) (+ skips); a store covers 4 rows x 16 consecutive frames -----------------
    if (!wave_active) return;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + j * 16 + l15;
            if (n >= a.ld_out) continue;
            const bool live = n < a.frames_out;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * 64 + i * 16 + kq * 4 + r;
                if (m >= a.c_out) continue;
                float v = relu_clamp((big[i][j][r] + small[i][j][r]) + a.bias[m]);
                const size_t off = (static_cast<size_t>(b) * a.c_out + m) * a.ld_out + n;
                if (a.s0) v += a.s0[off];
                if (a.s1) v += a.s1[off];
                if (a.s2) v += a.s2[off];
                a.y[off] = live ? v : 0.f;
            }
        }
    }
}

template <int S>
static int launch_packed(PackedConvArgs a, hipStream_t stream)
{
    using G = GeoP<S>;
    static const hipError_t attr0 = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_conv_bf16x3_kernel<S, false>),
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
    static const hipError_t attr1 = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_conv_bf16x3_kernel<S, true>),
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
    const hipError_t attr = attr0 != hipSuccess ? attr0 : attr1;
    if (attr != hipSuccess) {
        set_error("nbasr_dense_conv1d_fused_packed: cannot reserve %d bytes of LDS: %s", G::LDS_BYTES, hipGetErrorString(attr));
        return static_cast<int>(attr);
    }
    a.n_mt = (a.c_out + PB_M - 1) / PB_M;
    a.n_nt = (a.ld_out + PB_N - 1) / PB_N;
    const long long nwg = static_cast<long long>(a.n_mt) * a.n_nt * a.batch;
    NBASR_REQUIRE(nwg < (1ll << 31), NBASR_EINVAL, "nbasr_dense_conv1d_fused_packed: too many tiles (%lld)", nwg);
    if (a.ln_x.stats)
        hipLaunchKernelGGL((gemm_conv_bf16x3_kernel<S, true>), dim3(static_cast<unsigned>(nwg)), dim3(PB_THREADS), G::LDS_BYTES, stream, a);
    else
        hipLaunchKernelGGL((gemm_conv_bf16x3_kernel<S, false>), dim3(static_cast<unsigned>(nwg)), dim3(PB_THREADS), G::LDS_BYTES, stream, a);
    return launch_status("nbasr_dense_conv1d_fused_packed");
}

}  // namespace nbasr

using namespace nbasr;

extern "C" size_t nbasr_packed_dense_weights_bytes(int c_out, int c_in, int kernel)
{
    if (c_out <= 0 || c_in <= 0 || kernel != PB_TAPS) return 0;
    const size_t n_mt = (c_out + PB_M - 1) / PB_M, n_groups = (c_in + PB_CI - 1) / PB_CI;
    return n_mt * n_groups * PB_GROUP_BYTES;
}

extern "C" int nbasr_pack_dense_weights(const float* w, void* packed, int c_out, int c_in, int kernel, int stride,
                                        nbasr_stream_t stream)
{
    clear_error();
    NBASR_REQUIRE(c_out > 0 && c_in > 0, NBASR_EINVAL, "nbasr_pack_dense_weights: bad sizes");
    NBASR_REQUIRE(kernel == PB_TAPS && (stride == 1 || stride == 2), NBASR_EINVAL,
                  "nbasr_pack_dense_weights: (kernel=%d, stride=%d) unsupported (the downsample convs have k=8, s in {1,2})", kernel, stride);
    NBASR_REQUIRE(w && packed, NBASR_ENULL, "nbasr_pack_dense_weights: NULL pointer");
    NBASR_REQUIRE(aligned16(packed), NBASR_EALIGN, "nbasr_pack_dense_weights: packed buffer must be 16-byte aligned");
    const int n_mt = (c_out + PB_M - 1) / PB_M, n_groups = (c_in + PB_CI - 1) / PB_CI;
    hipLaunchKernelGGL(pack_dense_weights_kernel, dim3(2048), dim3(256), 0, as_stream(stream), w,
                       static_cast<__bf16*>(packed), c_out, c_in, n_mt, n_groups, pb_taps_per_step(stride));
    return launch_status("nbasr_pack_dense_weights");
}

static int dense_packed_impl(const float* x, const void* packed_w, const float* bias, const float* skip0,
                             const float* skip1, const float* skip2, float* y, int batch, int c_in,
                             int frames_in, int ld_in, int c_out, int ld_out, int kernel, int stride,
                             const nbasr_deferred_ln* ln, nbasr_stream_t stream)
{
    clear_error();
    NBASR_REQUIRE(batch >= 0 && c_in > 0 && c_out > 0 && frames_in >= 0, NBASR_EINVAL, "nbasr_dense_conv1d_fused_packed: bad sizes");
    NBASR_REQUIRE(kernel == PB_TAPS && (stride == 1 || stride == 2), NBASR_EINVAL,
                  "nbasr_dense_conv1d_fused_packed: (kernel=%d, stride=%d) unsupported; packed path covers k=8, s in {1,2}", kernel, stride);
    const int frames_out = (frames_in + stride - 1) / stride;
    NBASR_REQUIRE(ld_in >= frames_in, NBASR_EINVAL, "nbasr_dense_conv1d_fused_packed: ld_in=%d < frames_in=%d", ld_in, frames_in);
    NBASR_REQUIRE(ld_in % 4 == 0, NBASR_EALIGN, "nbasr_dense_conv1d_fused_packed: ld_in=%d must be a multiple of 4", ld_in);
    NBASR_REQUIRE(ld_out >= frames_out && ld_out % 4 == 0, NBASR_EALIGN,
                  "nbasr_dense_conv1d_fused_packed: ld_out=%d must be >= %d output frames and a multiple of 4", ld_out, frames_out);
    if (batch == 0 || frames_out == 0) return NBASR_OK;
    NBASR_REQUIRE(x && packed_w && bias && y, NBASR_ENULL, "nbasr_dense_conv1d_fused_packed: x, packed_w, bias, y must be non-NULL");
    NBASR_REQUIRE(aligned16(packed_w) && aligned16(x), NBASR_EALIGN, "nbasr_dense_conv1d_fused_packed: x and packed weights must be 16-byte aligned");
    PackedConvArgs a{};
    a.x = x; a.wp = static_cast<const unsigned char*>(packed_w); a.bias = bias; a.s0 = skip0; a.s1 = skip1; a.s2 = skip2; a.y = y;
    a.c_in = c_in; a.frames_in = frames_in; a.ld_in = ld_in; a.c_out = c_out; a.frames_out = frames_out; a.ld_out = ld_out;
    a.lpad = pad_left(kernel, 1, stride); a.n_groups = (c_in + PB_CI - 1) / PB_CI; a.batch = batch;
    NBASR_REQUIRE(!ln || (ln->stats && ln->gamma && ln->beta), NBASR_ENULL, "nbasr_dense_conv1d_fused_packed_ln: deferred LayerNorm needs stats, gamma and beta");
    a.ln_x = ln_ref(ln, true);
    return stride == 1 ? launch_packed<1>(a, as_stream(stream)) : launch_packed<2>(a, as_stream(stream));
}

extern "C" int nbasr_dense_conv1d_fused_packed(const float* x, const void* packed_w, const float* bias, const float* skip0,
                                               const float* skip1, const float* skip2, float* y, int batch, int c_in,
                                               int frames_in, int ld_in, int c_out, int ld_out, int kernel, int stride,
                                               nbasr_stream_t stream)
{
    return dense_packed_impl(x, packed_w, bias, skip0, skip1, skip2, y, batch, c_in, frames_in, ld_in, c_out, ld_out, kernel,
                             stride, nullptr, stream);
}

extern "C" int nbasr_dense_conv1d_fused_packed_ln(const float* x, const void* packed_w, const float* bias, float* y, int batch,
                                                  int c_in, int frames_in, int ld_in, int c_out, int ld_out, int kernel,
                                                  int stride, const nbasr_deferred_ln* ln, nbasr_stream_t stream)
{
    return dense_packed_impl(x, packed_w, bias, nullptr, nullptr, nullptr, y, batch, c_in, frames_in, ld_in, c_out, ld_out,
                             kernel, stride, ln, stream);
}
