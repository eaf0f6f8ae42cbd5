// A whole SearchCell whose three node operations are grouped convolutions, in ONE launch
// (reference model.py:49-59 SearchCell.forward over model.py:13-22 Node.forward and ops.py:24-30 PadConvRelu):
//     x1 = op0(x0n) + s00 x0n;   x2 = op1(x1) + s10 x0n + s11 x1;   x3 = op2(x2) + s20 x0n + s21 x1 + s22 x2
// with x0n = the cell input, optionally still carrying its LayerNorm (applied while loading, nbasr.h).
//
// Channel groups never mix inside such a cell, so one workgroup owns one (utterance, group) ROW for all frames and runs the
// three convolutions back to back; x1 and x2 never touch HBM -- they are exchanged between lanes through two LDS tiles
// (CG x frames floats each; a lane owns 4 frames, its k-tap window comes from its neighbours' chunks).  HBM traffic per cell
// drops from 6 tensor passes (+ skip re-reads) to 1 read + 1 write; the arithmetic is the per-node kernel's, in the same
// order, so the result is bit-identical to three nbasr_grouped_conv1d_fused launches.
//
// Algorithmic bytes credited per launch (bench.py): those of the three node operations it replaces.
#include "common.h"

#include <type_traits>

namespace nbasr {

template <int K, int D>
struct Win {
    static constexpr int LPAD = pad_left(K, D, 1);
    static constexpr int SPAN = (K - 1) * D;            // taps reach frames [t - LPAD, t - LPAD + SPAN]
    static constexpr int QL = (LPAD + 3) / 4;           // whole chunks left of the lane's own chunk
    static constexpr int QR = (SPAN - LPAD + 3) / 4;    // whole chunks right of it
    static constexpr int NCH = QL + 1 + QR;
    static constexpr int BASE = 4 * QL - LPAD;          // window index of (r = 0, tap = 0)
};

struct CellArgs {
    const float* x0; float* y;
    const float* w0; const float* w1; const float* w2;
    const float* b0; const float* b1; const float* b2;
    int channels, frames, ld, groups;
    int kd0, kd1, kd2;          // 0: k5 d1, 1: k5 d2, 2: k7 d1, 3: k7 d2
    int skips;                  // bit0 s00 | bit1 s10 | bit2 s11 | bit3 s20 | bit4 s21 | bit5 s22
    LnRef ln;                   // pending LayerNorm of x0 (stats == nullptr: x0 is already normalised)
};

// acc += conv over one group's CG input channels; `fetch(ci, c)` returns chunk (q - QL + c) of input channel ci (zeros outside)
template <int CG, int K, int D, class Fetch>
__device__ __forceinline__ void conv_accumulate(float (&acc)[CG][4], const float* __restrict__ wg, const float* __restrict__ bg,
                                                Fetch fetch)
{
    using W = Win<K, D>;
#pragma unroll
    for (int co = 0; co < CG; ++co) {
        const float bv = bg[co];
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[co][r] = bv;
    }
#pragma unroll 1
    for (int ci = 0; ci < CG; ++ci) {
        float xw[W::NCH * 4];
#pragma unroll
        for (int c = 0; c < W::NCH; ++c) {
            const float4 v = fetch(ci, c);
            xw[4 * c + 0] = v.x; xw[4 * c + 1] = v.y; xw[4 * c + 2] = v.z; xw[4 * c + 3] = v.w;
        }
#pragma unroll
        for (int j = 0; j < K; ++j) {
#pragma unroll
            for (int co = 0; co < CG; ++co) {
                const float wv = wg[(co * CG + ci) * K + j];
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    acc[co][r] = __builtin_fmaf(wv, xw[W::BASE + r + j * D], acc[co][r]);
            }
        }
    }
}

template <int CG>
__global__ __launch_bounds__(1024) void grouped_cell_kernel(const CellArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float tiles[];
    const int rl = blockDim.x * 4;                  // tile row length in floats (every lane owns one 16-byte chunk)
    float* const tA = tiles;                        // x1[co][frame]
    float* const tB = tiles + CG * rl;              // x2[co][frame]

    const int g = blockIdx.x, b = blockIdx.y;       // wave-uniform: weights / bias / gamma / beta come through s_load
    const int q = threadIdx.x;
    const int nq = a.ld >> 2, nql = blockDim.x;
    const bool in_row = q < nq;                     // lanes beyond the pitch only keep the tiles' tail at zero
    const int t0 = q * 4;
    const size_t row0 = (static_cast<size_t>(b) * a.channels + static_cast<size_t>(g) * CG) * a.ld;
    const float* __restrict__ x0 = a.x0 + row0;
    const bool has_ln = a.ln.stats != nullptr;
    const float4* __restrict__ mrow = has_ln ? reinterpret_cast<const float4*>(a.ln.stats + static_cast<size_t>(b) * 2 * a.ld) : nullptr;

    // own-chunk statistics (x0 as a skip input)
    float4 sm = make_float4(0.f, 0.f, 0.f, 0.f), sr = sm;
    if (has_ln && in_row) { sm = mrow[q]; sr = mrow[nq + q]; }
    auto x0n_own = [&](int co) -> float4 {          // the (normalised) cell input at this lane's frames, channel co
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (in_row) {
            v = *reinterpret_cast<const float4*>(x0 + static_cast<size_t>(co) * a.ld + t0);
            if (has_ln) {
                const float gam = a.ln.gamma[g * CG + co], bet = a.ln.beta[g * CG + co];
                v.x = ln_apply(v.x, sm.x, sr.x, gam, bet); v.y = ln_apply(v.y, sm.y, sr.y, gam, bet);
                v.z = ln_apply(v.z, sm.z, sr.z, gam, bet); v.w = ln_apply(v.w, sm.w, sr.w, gam, bet);
            }
        }
        return v;
    };
    auto mask_tail = [&](float (&o)[4]) {
#pragma unroll
        for (int r = 0; r < 4; ++r) if (!in_row || t0 + r >= a.frames) o[r] = 0.f;
    };

    float acc[CG][4];

    // ---- node 0: input from global memory (LayerNorm applied on load) ------------------------------------------------
    auto node0 = [&](auto kc, auto dc) {
        constexpr int K = decltype(kc)::value, D = decltype(dc)::value;
        using W = Win<K, D>;
        float mw[W::NCH * 4], rw[W::NCH * 4];
        if (has_ln) {
#pragma unroll
            for (int c = 0; c < W::NCH; ++c) {
                const int qq = q - W::QL + c;
                float4 m = make_float4(0.f, 0.f, 0.f, 0.f), r = m;
                if (qq >= 0 && qq < nq) { m = mrow[qq]; r = mrow[nq + qq]; }
                mw[4 * c + 0] = m.x; mw[4 * c + 1] = m.y; mw[4 * c + 2] = m.z; mw[4 * c + 3] = m.w;
                rw[4 * c + 0] = r.x; rw[4 * c + 1] = r.y; rw[4 * c + 2] = r.z; rw[4 * c + 3] = r.w;
            }
        }
        conv_accumulate<CG, K, D>(acc, a.w0 + static_cast<size_t>(g) * (CG * CG * K), a.b0 + g * CG, [&](int ci, int c) -> float4 {
            const int qq = q - W::QL + c;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (qq >= 0 && qq < nq) v = *reinterpret_cast<const float4*>(x0 + static_cast<size_t>(ci) * a.ld + 4 * qq);
            if (has_ln) {
                const float gam = a.ln.gamma[g * CG + ci], bet = a.ln.beta[g * CG + ci];
                v.x = ln_apply(v.x, mw[4 * c + 0], rw[4 * c + 0], gam, bet); v.y = ln_apply(v.y, mw[4 * c + 1], rw[4 * c + 1], gam, bet);
                v.z = ln_apply(v.z, mw[4 * c + 2], rw[4 * c + 2], gam, bet); v.w = ln_apply(v.w, mw[4 * c + 3], rw[4 * c + 3], gam, bet);
            }
            return v;
        });
    };
    // ---- nodes 1, 2: input from an LDS tile ------------------------------------------------------------------------------
    auto node_lds = [&](auto kc, auto dc, const float* tile, const float* w, const float* bias) {
        constexpr int K = decltype(kc)::value, D = decltype(dc)::value;
        using W = Win<K, D>;
        conv_accumulate<CG, K, D>(acc, w + static_cast<size_t>(g) * (CG * CG * K), bias + g * CG, [&](int ci, int c) -> float4 {
            const int qq = q - W::QL + c;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (qq >= 0 && qq < nql) v = *reinterpret_cast<const float4*>(tile + ci * rl + 4 * qq);
            return v;
        });
    };
    using I5 = std::integral_constant<int, 5>; using I7 = std::integral_constant<int, 7>;
    using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
#define NBASR_KD_SWITCH(kd, CALL)                                                          \
    switch (kd) {                                                                          \
        case 0: CALL(I5{}, I1{}); break; case 1: CALL(I5{}, I2{}); break;                  \
        case 2: CALL(I7{}, I1{}); break; default: CALL(I7{}, I2{}); break;                 \
    }

    // node 0 -> x1 -> tile A
#define NBASR_N0(kc, dc) node0(kc, dc)
    NBASR_KD_SWITCH(a.kd0, NBASR_N0)
#undef NBASR_N0
#pragma unroll
    for (int co = 0; co < CG; ++co) {
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = relu_clamp(acc[co][r]);
        if (a.skips & 1) { const float4 v = x0n_own(co); o[0] += v.x; o[1] += v.y; o[2] += v.z; o[3] += v.w; }
        mask_tail(o);
        *reinterpret_cast<float4*>(tA + co * rl + t0) = make_float4(o[0], o[1], o[2], o[3]);
    }
    __syncthreads();

    // node 1 -> x2 -> tile B
#define NBASR_N1(kc, dc) node_lds(kc, dc, tA, a.w1, a.b1)
    NBASR_KD_SWITCH(a.kd1, NBASR_N1)
#undef NBASR_N1
#pragma unroll
    for (int co = 0; co < CG; ++co) {
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = relu_clamp(acc[co][r]);
        if (a.skips & 2) { const float4 v = x0n_own(co); o[0] += v.x; o[1] += v.y; o[2] += v.z; o[3] += v.w; }
        if (a.skips & 4) { const float4 v = *reinterpret_cast<const float4*>(tA + co * rl + t0); o[0] += v.x; o[1] += v.y; o[2] += v.z; o[3] += v.w; }
        mask_tail(o);
        *reinterpret_cast<float4*>(tB + co * rl + t0) = make_float4(o[0], o[1], o[2], o[3]);
    }
    __syncthreads();

    // node 2 -> x3 -> global memory
#define NBASR_N2(kc, dc) node_lds(kc, dc, tB, a.w2, a.b2)
    NBASR_KD_SWITCH(a.kd2, NBASR_N2)
#undef NBASR_N2
#undef NBASR_KD_SWITCH
    if (!in_row) return;
#pragma unroll
    for (int co = 0; co < CG; ++co) {
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = relu_clamp(acc[co][r]);
        if (a.skips & 8) { const float4 v = x0n_own(co); o[0] += v.x; o[1] += v.y; o[2] += v.z; o[3] += v.w; }
        if (a.skips & 16) { const float4 v = *reinterpret_cast<const float4*>(tA + co * rl + t0); o[0] += v.x; o[1] += v.y; o[2] += v.z; o[3] += v.w; }
        if (a.skips & 32) { const float4 v = *reinterpret_cast<const float4*>(tB + co * rl + t0); o[0] += v.x; o[1] += v.y; o[2] += v.z; o[3] += v.w; }
        mask_tail(o);
        typedef float f4v __attribute__((ext_vector_type(4)));
        __builtin_nontemporal_store(f4v{o[0], o[1], o[2], o[3]}, reinterpret_cast<f4v*>(a.y + row0 + static_cast<size_t>(co) * a.ld + t0));
    }
}

template <int CG>
static int launch_cell(const CellArgs& a, int batch, hipStream_t stream)
{
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(grouped_cell_kernel<CG>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (attr != hipSuccess) {
        set_error("nbasr_grouped_cell_fused: cannot raise the dynamic LDS limit: %s", hipGetErrorString(attr));
        return static_cast<int>(attr);
    }
    const int nq = a.ld / 4;
    const int threads = ((nq + 63) / 64) * 64;
    const size_t lds = static_cast<size_t>(2) * CG * threads * 4 * sizeof(float);
    NBASR_REQUIRE(threads <= 1024 && lds <= 160 * 1024, NBASR_EINVAL,
                  "nbasr_grouped_cell_fused: a row of %d frames x %d channels per group does not fit one workgroup "
                  "(<= 4096 frames and 2*CG*frames*4 B <= 160 KiB); use the per-node launches", a.ld, CG);
    hipLaunchKernelGGL((grouped_cell_kernel<CG>), dim3(a.groups, batch), dim3(threads), lds, stream, a);
    return launch_status("nbasr_grouped_cell_fused");
}

static int kd_code(int kernel, int dilation)
{
    if (kernel == 5 && dilation == 1) return 0;
    if (kernel == 5 && dilation == 2) return 1;
    if (kernel == 7 && dilation == 1) return 2;
    if (kernel == 7 && dilation == 2) return 3;
    return -1;
}

}  // namespace nbasr

using namespace nbasr;

extern "C" int nbasr_grouped_cell_fits(int channels, int frames_ld, int groups)
{
    if (channels <= 0 || groups <= 0 || channels % groups || frames_ld <= 0 || frames_ld % 4) return 0;
    const int cg = channels / groups;
    if (cg != 6 && cg != 8 && cg != 10 && cg != 12) return 0;
    const int threads = ((frames_ld / 4 + 63) / 64) * 64;
    return threads <= 1024 && static_cast<size_t>(2) * cg * threads * 16 <= 160 * 1024;
}

extern "C" int nbasr_grouped_cell_fused(const float* x0, const float* w0, const float* b0, int k0, int d0,
                                        const float* w1, const float* b1, int k1, int d1,
                                        const float* w2, const float* b2, int k2, int d2, int skip_mask, float* y,
                                        int batch, int channels, int frames, int ld, int groups,
                                        const nbasr_deferred_ln* ln, nbasr_stream_t stream)
{
    clear_error();
    NBASR_REQUIRE(batch >= 0 && channels > 0 && frames >= 0 && groups > 0 && channels % groups == 0, NBASR_EINVAL,
                  "nbasr_grouped_cell_fused: bad sizes batch=%d channels=%d frames=%d groups=%d", batch, channels, frames, groups);
    if (batch == 0 || ld == 0) return NBASR_OK;
    NBASR_REQUIRE(x0 && w0 && b0 && w1 && b1 && w2 && b2 && y, NBASR_ENULL, "nbasr_grouped_cell_fused: NULL pointer");
    NBASR_REQUIRE(ld >= frames && ld % 4 == 0 && aligned16(x0) && aligned16(y), NBASR_EALIGN,
                  "nbasr_grouped_cell_fused: ld=%d must be >= frames=%d and a multiple of 4; x0, y 16-byte aligned", ld, frames);
    NBASR_REQUIRE(batch <= 65535 && skip_mask >= 0 && skip_mask < 64, NBASR_EINVAL, "nbasr_grouped_cell_fused: bad batch / skip mask");
    NBASR_REQUIRE(!ln || (ln->stats && ln->gamma && ln->beta && aligned16(ln->stats)), NBASR_ENULL,
                  "nbasr_grouped_cell_fused: deferred LayerNorm needs stats (16-byte aligned), gamma and beta");
    CellArgs a{};
    a.x0 = x0; a.y = y; a.w0 = w0; a.w1 = w1; a.w2 = w2; a.b0 = b0; a.b1 = b1; a.b2 = b2;
    a.channels = channels; a.frames = frames; a.ld = ld; a.groups = groups;
    a.kd0 = kd_code(k0, d0); a.kd1 = kd_code(k1, d1); a.kd2 = kd_code(k2, d2);
    NBASR_REQUIRE(a.kd0 >= 0 && a.kd1 >= 0 && a.kd2 >= 0, NBASR_EINVAL,
                  "nbasr_grouped_cell_fused: node ops must be conv5 / conv5d2 / conv7 / conv7d2 (got k=%d,%d,%d d=%d,%d,%d)", k0, k1, k2, d0, d1, d2);
    a.skips = skip_mask; a.ln = ln_ref(ln, true);
    hipStream_t s = as_stream(stream);
    switch (channels / groups) {
        case 6:  return launch_cell<6>(a, batch, s);
        case 8:  return launch_cell<8>(a, batch, s);
        case 10: return launch_cell<10>(a, batch, s);
        case 12: return launch_cell<12>(a, batch, s);
        default:
            set_error("nbasr_grouped_cell_fused: channels/groups=%d unsupported (model widths give 6, 8, 10, 12)", channels / groups);
            return NBASR_EINVAL;
    }
}
