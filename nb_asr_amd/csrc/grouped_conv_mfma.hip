// The grouped-convolution node op of the bf16 path ON THE MATRIX CORES (v_mfma_f32_16x16x32_bf16).
//
// Why: with bf16 rows the node kernel of grouped_conv_impl.h moves half the bytes for the same fp32 FMAs and sits at the vector
// ALU's pace (0.36 of the HBM peak in BASELINE configs[3]); the matrix cores have 16 x that rate to spare.
//
// Formulation -- no unaligned access, no im2col shuffling.  For one (utterance, group) and a tile of 128 output frames
//     F = F0 + 8 i + m,   i = 0..15 (row of a 16 x 16 MFMA tile),   m = 0..7,
//     out[co][F] = sum_{ci, j} w[co][ci][j] * x[ci][F - lpad + j d]      and with u = m - lpad + j d = 8 c + e:
//                = sum_{ci, c, e} W_m[(c, ci, e)][co] * x[ci][F0 + 8 (i + c) + e],     W_m[(c, ci, e)][co] = w[co][ci][(8 c + e - m + lpad) / d]
// (zero where that is not a tap).  So the A operand of every MFMA is ALIGNED 8-frame chunks of the input rows -- the same chunks
// for all eight m -- and the eight output phases m differ only in the weight matrix: the eight accumulators of a lane are then 8
// CONSECUTIVE frames of one output channel per register (one 16-byte store), and dilation is just another placement of the weights.  k-slots (8 k each) are (chunk c, input channel ci), chunk-major; a K-block of 32 k is four slots; a
// (phase m, K-block) pair whose slots hold no tap is skipped: 36-60 MFMAs per 128 frames x group (15-30 % of the matrix pipe's
// work is real) = ~1 000 cycles per tile against ~9 KB of HBM traffic: the kernel is HBM-bound with 4 x headroom.
//  * weights: the fragments B[(m, kb)][lane][8] are built once per weight version (pack_grouped_mfma_kernel) and copied into LDS
//    once per workgroup (36-60 KiB); the four waves of a workgroup work on the SAME group (different tiles) and share them;
//  * input: a wave stages its tile's rows (CG channels x (16 + 1 or 2) chunks) through a wave-private LDS buffer -- each chunk is
//    fetched once (the pending LayerNorm of the cell input is applied once per element on the way), then every A fragment is one
//    conflict-free ds_read_b128; the next tile's chunks are prefetched into registers under the MFMAs;
//  * epilogue: bias + relu + clamp in fp32, rounded to bf16 like the reference's op output, transposed through the staging buffer
//    so that skips are loaded and y is stored as whole 256-byte row segments; skip sum (LayerNorm on load for skip0) in fp32,
//    second rounding, streaming stores.
// The LayerNorm statistics of the output are NOT produced here (the four waves of a workgroup do not hold four groups of the same
// frames): the executor runs the statistics pass over the tensor instead (one extra read of 12).
#include "storage.h"

#include <cstdlib>
#include <type_traits>
#include <utility>

namespace nbasr {

// compile-time loop: f(std::integral_constant<int, 0>{}), f(<1>), ... -- the (phase, K-block) activity table must be a constant
// expression at every MFMA site (`if constexpr`), not something the optimiser may or may not fold
template <class F, int... I> __device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F> __device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int floor_div8(int v) { return v >= 0 ? v / 8 : -((-v + 7) / 8); }

template <int CG, int K, int D>
struct MGeo {
    static constexpr int LPAD = pad_left(K, D, 1), SPAN = (K - 1) * D;
    static constexpr int CMIN = floor_div8(-LPAD);                 // first chunk offset a tap can fall into (-1 or 0)
    static constexpr int CMAX = floor_div8(7 + SPAN - LPAD);       // last one (1)
    static constexpr int NC = CMAX - CMIN + 1;
    static constexpr int NS = NC * CG;                             // k-slots: (chunk, input channel), chunk-major
    static constexpr int NKB = (NS + 3) / 4;                       // K-blocks of four slots
    static constexpr int NCHK = 16 + NC - 1;                       // staged chunks per input row and tile
    static constexpr int STAGE_BYTES = CG * NCHK * 16;
    // tap index of (phase m, chunk offset c, element e), or -1
    static constexpr int tap(int m, int c, int e) {
        const int num = 8 * c + e - m + LPAD;
        return (num >= 0 && num % D == 0 && num / D < K) ? num / D : -1;
    }
    static constexpr bool active(int m, int kb) {
        for (int s = 4 * kb; s < 4 * kb + 4 && s < NS; ++s)
            for (int e = 0; e < 8; ++e)
                if (tap(m, CMIN + s / CG, e) >= 0) return true;
        return false;
    }
    static constexpr int pair_index(int m, int kb) {               // position of (m, kb) among the active pairs, phase-major
        int n = 0;
        for (int mm = 0; mm < 8; ++mm)
            for (int k = 0; k < NKB; ++k) {
                if (mm == m && k == kb) return n;
                if (active(mm, k)) ++n;
            }
        return n;
    }
    static constexpr int NPAIRS = pair_index(8, 0);
    static constexpr int FRAG_BYTES = NPAIRS * 1024;
};

// fragments of one layer: [group][pair][lane][8] bf16 from the fp32 weights (channels, CG, K)
template <int CG, int K, int D>
__global__ __launch_bounds__(256) void pack_grouped_mfma_kernel(const float* __restrict__ w, unsigned short* __restrict__ wp, int groups)
{
    using G = MGeo<CG, K, D>;
    const int total = groups * G::NPAIRS * 64 * 8;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int e = idx & 7, lane = (idx >> 3) & 63, pair = (idx >> 9) % G::NPAIRS, g = (idx >> 9) / G::NPAIRS;
        int m = 0, kb = 0, n = 0;
        bool found = false;
        for (int mm = 0; mm < 8 && !found; ++mm)
            for (int k = 0; k < G::NKB && !found; ++k)
                if (G::active(mm, k)) { if (n == pair) { m = mm; kb = k; found = true; } ++n; }
        const int co = lane & 15, s = 4 * kb + (lane >> 4);
        float v = 0.f;
        if (co < CG && s < G::NS) {
            const int j = G::tap(m, G::CMIN + s / CG, e);
            if (j >= 0) v = w[((static_cast<size_t>(g) * CG + co) * CG + (s % CG)) * K + j];
        }
        wp[idx] = __builtin_bit_cast(unsigned short, static_cast<__bf16>(v));
    }
}

struct MfmaArgs {
    const bf16_t* x; const unsigned char* wp; const float* bias;
    const bf16_t* s0; const bf16_t* s1; const bf16_t* s2; bf16_t* y;
    int batch, channels, frames, ld, groups, splits;
    LnRef ln_x, ln_s0;
};

// MW waves per workgroup share one copy of the group's weight fragments: 16 waves (one workgroup per CU at the widest group) keep
// four waves per SIMD in flight -- a tile is a chain of dependent round trips (stage -> fragments -> MFMAs -> skips -> store), and
// with 4 waves per workgroup (8-12 per CU, limited by the fragments' LDS) the first version ran at 9 us per tile and wave.
constexpr int MW = 16;

template <int CG, int K, int D, bool LNX>
__global__ __launch_bounds__(MW * 64) void grouped_conv_mfma_kernel(const MfmaArgs a)
{
    using G = MGeo<CG, K, D>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const frags = smem;                                          // [NPAIRS][64 lanes][16 B]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned char* const stage = smem + G::FRAG_BYTES + wave * G::STAGE_BYTES;    // wave-private: [CG][NCHK][16 B]
    const int g = blockIdx.x / a.splits, split = blockIdx.x - g * a.splits;
    const int i16 = lane & 15, kq = lane >> 4;
    // Tile row i computes the 8-frame chunk perm(i) = 4 (i & 3) + (i >> 2) of the tile (a 4 x 4 transpose of the row index): the
    // accumulator register r of lane (co, kq) -- tile row 4 kq + r -- is then chunk 4 r + kq, so ONE store / skip-load instruction
    // touches 64 contiguous bytes per channel row (the four kq lanes side by side) instead of four 16-byte pieces 64 bytes apart
    // (first version: every epilogue instruction was 48 separate quarter-sector requests: 2.3 TB/s at twice the vector kernel's time)
    const int prow = ((i16 & 3) << 2) | (i16 >> 2);

    // the group's weight fragments -> LDS (once per workgroup)
    {
        const u4v* __restrict__ src = reinterpret_cast<const u4v*>(a.wp + static_cast<size_t>(g) * G::FRAG_BYTES);
        u4v* dst = reinterpret_cast<u4v*>(frags);
        for (int i = threadIdx.x; i < G::FRAG_BYTES / 16; i += MW * 64) dst[i] = src[i];
    }
    __syncthreads();

    const int nq = a.ld >> 3;                            // 8-frame chunks per row
    const int tiles_per_row = (a.ld + 127) >> 7;
    const int n_tiles = a.batch * tiles_per_row;
    const int stride = a.splits * MW;
    constexpr int NLOAD = (CG * G::NCHK + 63) / 64;      // staged chunks per lane
    const float bias_v = i16 < CG ? a.bias[g * CG + i16] : 0.f;

    auto fetch = [&](int tile, u4v (&raw)[NLOAD]) {
        const int b = tile / tiles_per_row, tr = tile - b * tiles_per_row;
        const bf16_t* xg = a.x + (static_cast<size_t>(b) * a.channels + static_cast<size_t>(g) * CG) * a.ld;
#pragma unroll
        for (int n = 0; n < NLOAD; ++n) {
            const int it = lane + 64 * n;
            const int ch = it / G::NCHK, q = it - ch * G::NCHK;
            const int gq = tr * 16 + G::CMIN + q;
            raw[n] = u4v{0u, 0u, 0u, 0u};
            if (it < CG * G::NCHK && tile < n_tiles && gq >= 0 && gq < nq)
                raw[n] = *reinterpret_cast<const u4v*>(xg + static_cast<size_t>(ch) * a.ld + gq * 8);
        }
    };
    auto commit = [&](int tile, const u4v (&raw)[NLOAD]) {
        const int b = tile / tiles_per_row, tr = tile - b * tiles_per_row;
#pragma unroll
        for (int n = 0; n < NLOAD; ++n) {
            const int it = lane + 64 * n;
            if (it >= CG * G::NCHK) continue;
            u4v v = raw[n];
            if (LNX) {
                const int ch = it / G::NCHK, q = it - ch * G::NCHK;
                const int gq = tr * 16 + G::CMIN + q;
                if (gq >= 0 && gq < nq) {
                    const float* st = a.ln_x.stats + static_cast<size_t>(b) * 2 * a.ld + gq * 8;
                    float mu[8], rs[8];
                    load_frames<8>(st, mu);
                    load_frames<8>(st + a.ld, rs);
                    const float gam = a.ln_x.gamma[g * CG + ch], bet = a.ln_x.beta[g * CG + ch];
                    float f[8] = {bf16_lo(v.x), bf16_hi(v.x), bf16_lo(v.y), bf16_hi(v.y), bf16_lo(v.z), bf16_hi(v.z), bf16_lo(v.w), bf16_hi(v.w)};
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = ln_apply(f[e], mu[e], rs[e], gam, bet);
                    v = u4v{pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7])};
                }
            }
            *reinterpret_cast<u4v*>(stage + it * 16) = v;
        }
    };

    int tile = split * MW + wave;
    u4v raw[NLOAD];
    fetch(tile, raw);
    for (; tile < n_tiles; tile += stride) {
        commit(tile, raw);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // the wave's own staging writes have landed (wave-private buffer)
        const int next = tile + stride;
        fetch(next, raw);                                               // in flight under the MFMAs and the epilogue

        // K-block outer, phase inner: one A fragment (slot s = 4 kb + kq -> (chunk offset index, input channel); the 16 lanes of a
        // quarter read 256 contiguous bytes) is live at a time, the eight accumulators take turns -- independent MFMAs back to back
        floatx4 acc[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) acc[m] = floatx4{0.f, 0.f, 0.f, 0.f};
        static_for<G::NKB>([&](auto kb_) {
            constexpr int kb = decltype(kb_)::value;
            const int s = 4 * kb + kq;
            const int cidx = s / CG, ci = s - cidx * CG;
            u4v t = {0u, 0u, 0u, 0u};
            if (s < G::NS) t = *reinterpret_cast<const u4v*>(stage + (ci * G::NCHK + prow + cidx) * 16);
            const bf16x8 af = __builtin_bit_cast(bf16x8, t);
            static_for<8>([&](auto m_) {
                constexpr int m = decltype(m_)::value;
                if constexpr (G::active(m, kb)) {
                    constexpr int pair = G::pair_index(m, kb);
                    const bf16x8 bf = *reinterpret_cast<const bf16x8*>(frags + (pair * 64 + lane) * 16);
                    acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf, acc[m], 0, 0, 0);
                }
            });
        });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // every fragment read of this tile is done: the stage may be overwritten

        // epilogue, part 1: lane (co = i16, kq) holds the op's output z[co][F0 + 8 (4 r + kq) + m] in acc[m][r]; bias + relu + clamp,
        // rounded to bf16 (the reference rounds the op's output, too) and transposed through the wave's staging buffer
        // ([CG][128 frames] bf16 = CG x 256 bytes, no larger than the input stage) ...
        static_assert(CG * 256 <= G::STAGE_BYTES, "the output tile must fit the staging buffer");
        if (i16 < CG) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float o[8];
#pragma unroll
                for (int m = 0; m < 8; ++m) o[m] = relu_clamp(acc[m][r] + bias_v);
                const u4v t = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7])};
                *reinterpret_cast<u4v*>(stage + i16 * 256 + (4 * r + kq) * 16) = t;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // ... part 2, ROW-COALESCED: 16 lanes x 16 bytes = one whole 256-byte row segment of a channel, four channels per
        // instruction -- skips are loaded, and y is stored, as full cache lines (the accumulator layout itself gives 64-byte pieces:
        // at 2.3-2.8 TB/s the first versions of this kernel were slower than the vector-ALU one)
        {
            const int b = tile / tiles_per_row, tr = tile - b * tiles_per_row;
            const int f0 = tr * 128 + 8 * i16;
            const int rsub = lane >> 4;
            float mu[8], rs[8];
            const bool ln0 = a.s0 && a.ln_s0.stats;
            if (ln0 && f0 < a.ld) {
                const float* st = a.ln_s0.stats + static_cast<size_t>(b) * 2 * a.ld + f0;
                load_frames<8>(st, mu);
                load_frames<8>(st + a.ld, rs);
            }
#pragma unroll
            for (int rg = 0; rg < (CG + 3) / 4; ++rg) {
                const int co = rg * 4 + rsub;
                if (co >= CG || f0 >= a.ld) continue;
                const size_t row = (static_cast<size_t>(b) * a.channels + static_cast<size_t>(g) * CG + co) * a.ld;
                float o[8];
                load_frames<8>(reinterpret_cast<const bf16_t*>(stage + co * 256 + i16 * 16), o);
                if (a.s0) {
                    float v[8];
                    load_frames<8>(a.s0 + row + f0, v);
                    if (ln0) {
                        const float gam0 = a.ln_s0.gamma[g * CG + co], bet0 = a.ln_s0.beta[g * CG + co];
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = ln_apply(v[e], mu[e], rs[e], gam0, bet0);
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] += v[e];
                }
                if (a.s1) {
                    float v[8];
                    load_frames<8>(a.s1 + row + f0, v);
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] += v[e];
                }
                if (a.s2) {
                    float v[8];
                    load_frames<8>(a.s2 + row + f0, v);
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] += v[e];
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) if (f0 + e >= a.frames) o[e] = 0.f;      // pitch columns stay zero
                store_frames<8, true>(a.y + row + f0, o);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // the output tile has been read back: the stage is free for the next input
    }
}

template <int CG, int K, int D>
static int launch_mfma(const MfmaArgs& a, hipStream_t stream)
{
    using G = MGeo<CG, K, D>;
    constexpr int LDS = G::FRAG_BYTES + MW * G::STAGE_BYTES;
    static_assert(LDS <= 160 * 1024, "weight fragments + staging must fit the LDS");
    static const hipError_t attr0 = hipFuncSetAttribute(reinterpret_cast<const void*>(grouped_conv_mfma_kernel<CG, K, D, false>),
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    static const hipError_t attr1 = hipFuncSetAttribute(reinterpret_cast<const void*>(grouped_conv_mfma_kernel<CG, K, D, true>),
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (attr0 != hipSuccess || attr1 != hipSuccess) {
        set_error("nbasr_grouped_conv1d_node_mfma: cannot reserve %d bytes of LDS", LDS);
        return static_cast<int>(attr0 != hipSuccess ? attr0 : attr1);
    }
    const dim3 grid(a.groups * a.splits);
    if (a.ln_x.stats) hipLaunchKernelGGL((grouped_conv_mfma_kernel<CG, K, D, true>), grid, dim3(MW * 64), LDS, stream, a);
    else hipLaunchKernelGGL((grouped_conv_mfma_kernel<CG, K, D, false>), grid, dim3(MW * 64), LDS, stream, a);
    return launch_status("nbasr_grouped_conv1d_node_mfma");
}

template <int CG, int K, int D>
static size_t frag_bytes() { return MGeo<CG, K, D>::FRAG_BYTES; }

#define NBASR_MFMA_DISPATCH(FN, cg, k, d, ...)                                                            \
    do {                                                                                                  \
        const int key_ = (cg) * 100 + (k) * 10 + (d);                                                     \
        switch (key_) {                                                                                   \
            case 651: return FN<6, 5, 1>(__VA_ARGS__);   case 652: return FN<6, 5, 2>(__VA_ARGS__);      \
            case 671: return FN<6, 7, 1>(__VA_ARGS__);   case 672: return FN<6, 7, 2>(__VA_ARGS__);      \
            case 851: return FN<8, 5, 1>(__VA_ARGS__);   case 852: return FN<8, 5, 2>(__VA_ARGS__);      \
            case 871: return FN<8, 7, 1>(__VA_ARGS__);   case 872: return FN<8, 7, 2>(__VA_ARGS__);      \
            case 1051: return FN<10, 5, 1>(__VA_ARGS__); case 1052: return FN<10, 5, 2>(__VA_ARGS__);    \
            case 1071: return FN<10, 7, 1>(__VA_ARGS__); case 1072: return FN<10, 7, 2>(__VA_ARGS__);    \
            case 1251: return FN<12, 5, 1>(__VA_ARGS__); case 1252: return FN<12, 5, 2>(__VA_ARGS__);    \
            case 1271: return FN<12, 7, 1>(__VA_ARGS__); case 1272: return FN<12, 7, 2>(__VA_ARGS__);    \
            default: break;                                                                               \
        }                                                                                                 \
    } while (0)

template <int CG, int K, int D>
static int pack_mfma(const float* w, void* packed, int groups, hipStream_t stream)
{
    using G = MGeo<CG, K, D>;
    const int total = groups * G::NPAIRS * 512;
    hipLaunchKernelGGL((pack_grouped_mfma_kernel<CG, K, D>), dim3((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096), dim3(256), 0, stream,
                       w, static_cast<unsigned short*>(packed), groups);
    return launch_status("nbasr_pack_grouped_weights_mfma");
}

static size_t frag_bytes_rt(int cg, int k, int d) { NBASR_MFMA_DISPATCH(frag_bytes, cg, k, d); return 0; }
static int pack_rt(int cg, int k, int d, const float* w, void* p, int groups, hipStream_t s) { NBASR_MFMA_DISPATCH(pack_mfma, cg, k, d, w, p, groups, s); return NBASR_EINVAL; }
static int launch_rt(int cg, int k, int d, const MfmaArgs& a, hipStream_t s) { NBASR_MFMA_DISPATCH(launch_mfma, cg, k, d, a, s); return NBASR_EINVAL; }

}  // namespace nbasr

using namespace nbasr;

extern "C" size_t nbasr_grouped_mfma_weights_bytes(int channels, int groups, int kernel, int dilation)
{
    if (channels <= 0 || groups <= 0 || channels % groups) return 0;
    return frag_bytes_rt(channels / groups, kernel, dilation) * static_cast<size_t>(groups);
}

extern "C" int nbasr_pack_grouped_weights_mfma(const float* w, void* packed, int channels, int groups, int kernel, int dilation,
                                               nbasr_stream_t stream)
{
    clear_error();
    NBASR_REQUIRE(w && packed, NBASR_ENULL, "nbasr_pack_grouped_weights_mfma: NULL pointer");
    NBASR_REQUIRE(nbasr_grouped_mfma_weights_bytes(channels, groups, kernel, dilation) != 0, NBASR_EINVAL,
                  "nbasr_pack_grouped_weights_mfma: unsupported (channels=%d, groups=%d, kernel=%d, dilation=%d)", channels, groups, kernel, dilation);
    return pack_rt(channels / groups, kernel, dilation, w, packed, groups, as_stream(stream));
}

extern "C" int nbasr_grouped_conv1d_node_mfma(const void* x, const void* packed_w, const float* bias, const void* skip0, const void* skip1,
                                              const void* skip2, void* y, int batch, int channels, int frames, int ld, int groups,
                                              int kernel, int dilation, const nbasr_deferred_ln* ln, int ln_on_x, int ln_on_skip0,
                                              nbasr_stream_t stream)
{
    clear_error();
    NBASR_REQUIRE(batch >= 0 && channels > 0 && frames >= 0 && groups > 0 && channels % groups == 0, NBASR_EINVAL,
                  "nbasr_grouped_conv1d_node_mfma: bad sizes");
    if (batch == 0 || ld == 0) return NBASR_OK;
    NBASR_REQUIRE(x && packed_w && bias && y, NBASR_ENULL, "nbasr_grouped_conv1d_node_mfma: x, packed_w, bias, y must be non-NULL");
    NBASR_REQUIRE(ld >= frames && ld % 8 == 0, NBASR_EALIGN, "nbasr_grouped_conv1d_node_mfma: ld=%d must be >= frames=%d and a multiple of 8", ld, frames);
    NBASR_REQUIRE(aligned16(x) && aligned16(y) && aligned16(skip0) && aligned16(skip1) && aligned16(skip2) && aligned16(packed_w), NBASR_EALIGN,
                  "nbasr_grouped_conv1d_node_mfma: pointers must be 16-byte aligned");
    NBASR_REQUIRE(nbasr_grouped_mfma_weights_bytes(channels, groups, kernel, dilation) != 0, NBASR_EINVAL,
                  "nbasr_grouped_conv1d_node_mfma: unsupported (channels/groups=%d, kernel=%d, dilation=%d)", channels / groups, kernel, dilation);
    const bool any_ln = ln && (ln_on_x || (ln_on_skip0 && skip0));
    NBASR_REQUIRE(!any_ln || (ln->stats && ln->gamma && ln->beta && aligned16(ln->stats)), NBASR_ENULL,
                  "nbasr_grouped_conv1d_node_mfma: deferred LayerNorm needs stats (16-byte aligned), gamma and beta");
    MfmaArgs a{};
    a.x = static_cast<const bf16_t*>(x); a.wp = static_cast<const unsigned char*>(packed_w); a.bias = bias;
    a.s0 = static_cast<const bf16_t*>(skip0); a.s1 = static_cast<const bf16_t*>(skip1); a.s2 = static_cast<const bf16_t*>(skip2);
    a.y = static_cast<bf16_t*>(y);
    a.batch = batch; a.channels = channels; a.frames = frames; a.ld = ld; a.groups = groups;
    a.ln_x = ln_ref(ln, ln_on_x != 0); a.ln_s0 = ln_ref(ln, ln_on_skip0 != 0 && skip0 != nullptr);
    // workgroups: `splits` per group, each of its 16 waves strides over the (utterance, 128-frame tile) list; ~2 workgroups per CU
    // in all (one resident at a time at the widest group: its fragments take 60 of the 160 KiB), never more waves than tiles
    const long long tiles = static_cast<long long>(batch) * ((ld + 127) / 128);
    // whole rounds matter: one workgroup per CU is resident, and 600 workgroups (2.3 rounds of 256) ran 10 % slower than 500 (1.95)
    int splits = 512 / groups;
    if (splits < 1) splits = 1;
    if (const char* env = getenv("NBASR_MFMA_SPLITS")) { const int v = atoi(env); if (v > 0) splits = v; }      // diagnostics
    if (static_cast<long long>(splits) * MW > tiles) splits = static_cast<int>((tiles + MW - 1) / MW);
    a.splits = splits < 1 ? 1 : splits;
    return launch_rt(channels / groups, kernel, dilation, a, as_stream(stream));
}
