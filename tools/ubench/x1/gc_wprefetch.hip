// Exploration for the next round (not part of the library): the pipelined, output-split fp32 node kernel of
// nb_asr_amd/csrc/grouped_conv_osplit.hip reduced to its plain flavour (no LayerNorm on load, no skips, no statistics), with one
// switch: WPF = the scalar weight loads of channel ci + 1 are issued BEFORE the FMAs of channel ci, into a second register set
// (2 x CO x K SGPRs; fits only with the output split: 60 at CG = 12), the way the window loads already are.  DESIGN.md section 7.1 names
// the weight loads -- waited for in full before a channel's first FMA -- as the next serial piece of a wave's loop.
// GEN = the library kernel's generic prologue / epilogue around the same loop (three optional skip inputs, a deferred LayerNorm on the
// first, the ragged-tail vote): what does generality cost a wave that lives for 12 channel iterations (block 3)?
// NS = the alternative: the epilogue instantiated per number of skips, branch-free (hipcc puts a full vmcnt(0) at every control-flow
// join, and stores count in vmcnt: the generic epilogue's per-channel `if (s0)` serialises its own stores).
//   hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -shared -I nb_asr_amd/csrc -I include -x hip tools/ubench/x1/gc_wprefetch.hip -o tools/ubench/x1/libgc_wpf.so
#include <hip/hip_runtime.h>
#include "common.h"

#ifndef X1_SEPARATE_STAGES
#define X1_SEPARATE_STAGES 0
#endif

namespace x1 {
using nbasr::pad_left;
using nbasr::relu_clamp;
using nbasr::LnRef;
using nbasr::ln_apply;

typedef float gc_f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float4 row_chunk(const float* row, int row_bytes, int byte_offset)
{
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(row), 0, row_bytes, 0x00020000);
    const gc_f4 f = __builtin_bit_cast(gc_f4, __builtin_amdgcn_raw_buffer_load_b128(rs, byte_offset, 0, 0));
    return make_float4(f[0], f[1], f[2], f[3]);
}

template <int CG, int K, int D, bool WPF, bool GEN, int NS = 0>
__global__ __launch_bounds__(256) void node_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                   const float* __restrict__ s0, const float* __restrict__ s1, const float* __restrict__ s2,
                                                   float* __restrict__ y, int channels, int frames, int ld, int groups, const LnRef ln_s0)
{
    constexpr int OS = 2, CO = CG / OS, GPW = 4 / OS;
    constexpr int LPAD = pad_left(K, D, 1), SPAN = (K - 1) * D, QL = (LPAD + 3) / 4, QR = (SPAN - LPAD + 3) / 4, NCH = QL + 1 + QR, BASE = 4 * QL - LPAD;
    const int nq = ld >> 2, lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = blockIdx.x * 64 + lane;
    const int g = __builtin_amdgcn_readfirstlane(blockIdx.y * GPW + wave / OS);
    const int co0 = (wave % OS) * CO, b = blockIdx.z;
    if (g >= groups) return;
    const size_t row0 = (static_cast<size_t>(b) * channels + static_cast<size_t>(g) * CG) * ld;
    const float* __restrict__ wg = w + (static_cast<size_t>(g) * CG + co0) * (CG * K);
    const float* __restrict__ bg = bias + g * CG + co0;
    const int row_bytes = ld * 4, off0 = (q - QL) * 16;
    float acc[CO][4];
#pragma unroll
    for (int co = 0; co < CO; ++co) {
        const float bv = bg[co];
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[co][r] = bv;
    }
    const float* __restrict__ xg = x + row0;
    auto fetch = [&](int ci, float4 (&dst)[NCH]) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) dst[c] = row_chunk(xg + static_cast<size_t>(ci) * ld, row_bytes, off0 + 16 * c);
    };
    auto weights = [&](int ci, float (&dst)[CO * K]) {
#pragma unroll
        for (int co = 0; co < CO; ++co)
#pragma unroll
            for (int j = 0; j < K; ++j) dst[co * K + j] = wg[(co * CG + ci) * K + j];
    };
    auto consume = [&](const float (&wv)[CO * K], const float4 (&src)[NCH]) {
        float xw[NCH * 4];
#pragma unroll
        for (int c = 0; c < NCH; ++c) { xw[4 * c + 0] = src[c].x; xw[4 * c + 1] = src[c].y; xw[4 * c + 2] = src[c].z; xw[4 * c + 3] = src[c].w; }
#pragma unroll
        for (int j = 0; j < K; ++j)
#pragma unroll
            for (int co = 0; co < CO; ++co)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[co][r] = __builtin_fmaf(wv[co * K + j], xw[BASE + r + j * D], acc[co][r]);
    };
    float4 wa[NCH], wb[NCH];
    float w0[CO * K], w1[CO * K];
    fetch(0, wa);
    if (WPF) weights(0, w0);
#pragma unroll 1
    for (int ci = 0; ci < CG; ci += 2) {
        const int nxt = ci + 2 < CG ? ci + 2 : CG - 1;          // last round: a redundant reload instead of a branch
        fetch(ci + 1, wb);
        if (WPF) weights(ci + 1, w1); else weights(ci, w0);
        consume(w0, wa);
        fetch(nxt, wa);
        if (WPF) weights(nxt, w0); else weights(ci + 1, w1);
        consume(w1, wb);
    }
    if (q >= nq) return;
    const int t0 = q * 4;
    if constexpr (GEN) {
        const bool ragged = __any(t0 + 3 >= frames) != 0;
        float4 sm = make_float4(0.f, 0.f, 0.f, 0.f), sr = sm;
        if (s0 && ln_s0.stats) {
            const float4* __restrict__ mrow = reinterpret_cast<const float4*>(ln_s0.stats + static_cast<size_t>(b) * 2 * ld);
            sm = mrow[q];
            sr = mrow[nq + q];
        }
#pragma unroll
        for (int co = 0; co < CO; ++co) {
            const size_t off = row0 + static_cast<size_t>(co0 + co) * ld + t0;
            float o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = relu_clamp(acc[co][r]);
            if (s0) {
                float4 v = *reinterpret_cast<const float4*>(s0 + off);
                if (ln_s0.stats) {
                    const float gam = ln_s0.gamma[g * CG + co0 + co], bet = ln_s0.beta[g * CG + co0 + co];
                    v.x = ln_apply(v.x, sm.x, sr.x, gam, bet); v.y = ln_apply(v.y, sm.y, sr.y, gam, bet);
                    v.z = ln_apply(v.z, sm.z, sr.z, gam, bet); v.w = ln_apply(v.w, sm.w, sr.w, gam, bet);
                }
                o[0] += v.x; o[1] += v.y; o[2] += v.z; o[3] += v.w;
            }
            if (s1) { const float4 v = *reinterpret_cast<const float4*>(s1 + off); o[0] += v.x; o[1] += v.y; o[2] += v.z; o[3] += v.w; }
            if (s2) { const float4 v = *reinterpret_cast<const float4*>(s2 + off); o[0] += v.x; o[1] += v.y; o[2] += v.z; o[3] += v.w; }
            if (ragged) {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (t0 + r >= frames) o[r] = 0.f;
            }
            typedef float f4v __attribute__((ext_vector_type(4)));
            __builtin_nontemporal_store(f4v{o[0], o[1], o[2], o[3]}, reinterpret_cast<f4v*>(y + off));
        }
        return;
    }
    // branch-free in the skip dimension (NS is a template parameter): every skip load is requested before the first store, the ragged tail
    // is a select -- no control-flow join, so no vmcnt(0) between the stores
    float4 sk[NS > 0 ? NS : 1][CO];
#pragma unroll
    for (int co = 0; co < CO; ++co) {
        const size_t off = row0 + static_cast<size_t>(co0 + co) * ld + t0;
        if (NS > 0) sk[0][co] = *reinterpret_cast<const float4*>(s0 + off);
        if (NS > 1) sk[1 % (NS > 0 ? NS : 1)][co] = *reinterpret_cast<const float4*>(s1 + off);
    }
#pragma unroll
    for (int co = 0; co < CO; ++co) {
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = relu_clamp(acc[co][r]);
        if (NS > 0) { o[0] += sk[0][co].x; o[1] += sk[0][co].y; o[2] += sk[0][co].z; o[3] += sk[0][co].w; }
        if (NS > 1) { const float4 v = sk[1 % (NS > 0 ? NS : 1)][co]; o[0] += v.x; o[1] += v.y; o[2] += v.z; o[3] += v.w; }
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = t0 + r < frames ? o[r] : 0.f;
        typedef float f4v __attribute__((ext_vector_type(4)));
        __builtin_nontemporal_store(f4v{o[0], o[1], o[2], o[3]}, reinterpret_cast<f4v*>(y + row0 + static_cast<size_t>(co0 + co) * ld + t0));
    }
}

// COOP: the two waves of an output-split group stop loading the same windows.  Wave h of the pair requests channel 2 s + h of stage s --
// ONE 16-byte chunk per lane (plus the two halo chunks of the 64-chunk block, lanes 0 and 1; every other lane's halo offset is out of
// range, i.e. a zero from the bounds check and no memory request) -- writes it to LDS, and both waves read their three chunks per channel
// from there.  Distinct bytes in flight per wave double (2 stages x 1 KiB against 2 windows x 3 KiB of which a third is distinct and half
// of that shared with the partner wave), L1 / TA requests drop 6 x.  One workgroup barrier per stage, LDS double-buffered.
template <int CG, int K, int D>
__global__ __launch_bounds__(256) void node_coop_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                        float* __restrict__ y, int channels, int frames, int ld, int groups)
{
    static_assert(K == 5 && D == 1 && CG % 4 == 2 || CG % 4 == 0, "window = chunks q - 1, q, q + 1; stages unrolled by two where CG / 2 is even");
    constexpr int CO = CG / 2, LPAD = pad_left(K, D, 1), BASE = 4 - LPAD, SLOTS = 67, NST = CG / 2;
    __shared__ float4 s_win[2][2][2][SLOTS];                    // [stage buffer][group slot][channel of the stage][chunk slot]
    const int nq = ld >> 2, lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int slot = wave >> 1, h = wave & 1;
    const int q0 = blockIdx.x * 64, q = q0 + lane;
    const int g_raw = __builtin_amdgcn_readfirstlane(blockIdx.y * 2 + slot);
    const int g = g_raw < groups ? g_raw : groups - 1;
    const int co0 = h * CO, b = blockIdx.z;
    const size_t row0 = (static_cast<size_t>(b) * channels + static_cast<size_t>(g) * CG) * ld;
    // weights re-laid-out by the caller: [group][half][stage][co][channel of the stage][tap] -- a stage's 2 x CO x K scalars contiguous
    const float* __restrict__ wg = w + (static_cast<size_t>(g) * 2 + h) * (NST * CO * 2 * K);
    const float* __restrict__ bg = bias + g * CG + co0;
    const int row_bytes = ld * 4;
    const int own_off = q * 16;
    const int halo_off = lane == 0 ? (q0 - 1) * 16 : lane == 1 ? (q0 + 64) * 16 : 0x7ffffff0;
    const int halo_slot = lane == 0 ? 0 : lane == 1 ? 65 : 66;
    float acc[CO][4];
#pragma unroll
    for (int co = 0; co < CO; ++co) {
        const float bv = bg[co];
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[co][r] = bv;
    }
    const float* __restrict__ xg = x + row0;
    struct Stage { float4 own, halo; };
    auto request = [&](int s) {
        const int ci = 2 * (s < NST ? s : NST - 1) + h;          // past the end: a redundant reload instead of a branch
        const float* row = xg + static_cast<size_t>(ci) * ld;
        return Stage{row_chunk(row, row_bytes, own_off), row_chunk(row, row_bytes, halo_off)};
    };
    auto publish = [&](int s, const Stage& st) {
        s_win[s & 1][slot][h][lane + 1] = st.own;
        s_win[s & 1][slot][h][halo_slot] = st.halo;
    };
    auto consume = [&](int s) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            __builtin_amdgcn_sched_barrier(0);                  // one channel at a time: without it hipcc hoists both channels' weights
            int sv = s;                                         // pinned here: hipcc otherwise hoists the NEXT stage's 2 x CO x K scalars
            asm volatile("" : "+s"(sv));                         // above this stage's FMAs (120 SGPRs at CG = 12 -> spills)
            const float* __restrict__ ws = wg + static_cast<size_t>(sv) * (CO * 2 * K);
            float xw[12];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float4 v = s_win[s & 1][slot][c][lane + k];
                xw[4 * k + 0] = v.x; xw[4 * k + 1] = v.y; xw[4 * k + 2] = v.z; xw[4 * k + 3] = v.w;
            }
#pragma unroll
            for (int j = 0; j < K; ++j)
#pragma unroll
                for (int co = 0; co < CO; ++co) {
                    const float wv = ws[(co * 2 + c) * K + j];
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[co][r] = __builtin_fmaf(wv, xw[BASE + r + j * D], acc[co][r]);
                }
        }
    };
    Stage ra = request(0), rb = request(1);
    int nst = NST;
#if X1_SEPARATE_STAGES
    // opaque stage count: the inner branch then stays, and with it a control-flow join (vmcnt(0)) between the two stages of a pass.  60-72
    // VGPRs instead of 62-95 and no scalar spills -- but only ONE stage of lookahead: measured 0.97 / 1.01 / 1.00 / 1.05 x the pipelined
    // output-split kernel on blocks 0-3, against 0.97 / 0.93 / 0.92-0.94 / 1.07-1.11 x with the stages interleaved (default): the depth of
    // the lookahead is what pays, not the occupancy.
    asm volatile("" : "+s"(nst));
#endif
#pragma unroll 1
    for (int s = 0; s < nst; s += 2) {
        publish(s, ra);
        __syncthreads();
        ra = request(s + 2);
        __builtin_amdgcn_sched_barrier(0);
        consume(s);
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < nst) {                                      // (uniform; CG / 2 odd: the last pass has one stage)
            publish(s + 1, rb);
            __syncthreads();
            rb = request(s + 3);
            __builtin_amdgcn_sched_barrier(0);
            consume(s + 1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (q >= nq || g_raw >= groups) return;
    const int t0 = q * 4;
#pragma unroll
    for (int co = 0; co < CO; ++co) {
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = t0 + r < frames ? relu_clamp(acc[co][r]) : 0.f;
        typedef float f4v __attribute__((ext_vector_type(4)));
        __builtin_nontemporal_store(f4v{o[0], o[1], o[2], o[3]}, reinterpret_cast<f4v*>(y + row0 + static_cast<size_t>(co0 + co) * ld + t0));
    }
}

template <int CG>
static int launch(int mode, const float* x, const float* w, const float* bias, const float* s0, const float* s1, float* y, int batch, int channels,
                  int frames, int ld, hipStream_t s)
{
    const int groups = channels / CG, nq = ld >> 2;
    const dim3 grid((nq + 63) / 64, (groups + 1) / 2, batch);
    const float* none = nullptr;
    const LnRef no_ln{nullptr, nullptr, nullptr};
#define X1_LAUNCH(...) hipLaunchKernelGGL((node_kernel<CG, 5, 1, __VA_ARGS__>), grid, dim3(256), 0, s, x, w, bias, s0, s1, none, y, channels, frames, ld, groups, no_ln)
    if (mode == 1)      X1_LAUNCH(true, false);
    else if (mode == 2) X1_LAUNCH(false, true);                 // generic epilogue: s0 / s1 may be NULL
    else if (mode == 4) hipLaunchKernelGGL((node_coop_kernel<CG, 5, 1>), grid, dim3(256), 0, s, x, w, bias, y, channels, frames, ld, groups);
    else if (mode == 3) { if (s1) X1_LAUNCH(false, false, 2); else if (s0) X1_LAUNCH(false, false, 1); else X1_LAUNCH(false, false, 0); }
    else                X1_LAUNCH(false, false);
#undef X1_LAUNCH
    return static_cast<int>(hipGetLastError());
}
}  // namespace x1

// mode: 0 = base (no skips), 1 = weight prefetch (no skips), 2 = base loop inside the generic prologue / epilogue (skips by pointer),
//       3 = base loop with the branch-free epilogue instantiated for the number of non-NULL skips, 4 = cooperative window loads (no skips)
extern "C" int x1_node(int mode, const float* x, const float* w, const float* bias, const float* s0, const float* s1, float* y, int batch,
                       int channels, int frames, int ld, void* stream)
{
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (channels / 100) {
        case 6:  return x1::launch<6>(mode, x, w, bias, s0, s1, y, batch, channels, frames, ld, s);
        case 8:  return x1::launch<8>(mode, x, w, bias, s0, s1, y, batch, channels, frames, ld, s);
        case 10: return x1::launch<10>(mode, x, w, bias, s0, s1, y, batch, channels, frames, ld, s);
        case 12: return x1::launch<12>(mode, x, w, bias, s0, s1, y, batch, channels, frames, ld, s);
    }
    return -1;
}
