// Storage-type helpers shared by the kernels that exist for both activation precisions:
//   float  : the fp32 path (BASELINE configs 1-3)
//   bf16_t : the bf16 path (BASELINE config 4): activations and GEMM operands stored as bfloat16, every sum, LayerNorm
//            statistic and LSTM state in fp32, ONE rounding to bf16 when a tensor is written back to HBM.
// A "chunk" is 16 bytes of one activation row: 4 fp32 frames or 8 bf16 frames; rows are pitched to whole chunks.
#pragma once
#include "common.h"

namespace nbasr {

struct bf16_t { unsigned short bits; };          // raw bfloat16 (no arithmetic; converted on load / store)

template <typename T> struct Chunk;
template <> struct Chunk<float>  { static constexpr int FR = 4; };
template <> struct Chunk<bf16_t> { static constexpr int FR = 8; };

typedef float f2v __attribute__((ext_vector_type(2)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef unsigned u2v __attribute__((ext_vector_type(2)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf2v __attribute__((ext_vector_type(2)));

// two floats -> one dword of two bfloat16 (round to nearest even; NaN stays NaN: v_cvt_pk_bf16_f32)
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f2v{lo, hi}, bf2v));
}
__device__ __forceinline__ float bf16_lo(unsigned d) { return __uint_as_float(d << 16); }
__device__ __forceinline__ float bf16_hi(unsigned d) { return __uint_as_float(d & 0xffff0000u); }

// N consecutive frames (N = 4 or 8) of a row -> fp32 registers.  The pointer must be aligned to the access it implies:
// fp32 16 B per 4 frames; bf16 8 B per 4 frames, 16 B per 8 frames.
template <int N> __device__ __forceinline__ void load_frames(const float* __restrict__ p, float (&v)[N]) {
    static_assert(N == 2 || N == 4 || N == 8, "2, 4 or 8 frames per access");
    if constexpr (N == 2) {
        const f2v t = *reinterpret_cast<const f2v*>(p);
        v[0] = t.x; v[1] = t.y;
        return;
    }
#pragma unroll
    for (int h = 0; h < N / 4; ++h) {
        const f4v t = *reinterpret_cast<const f4v*>(p + 4 * h);
        v[4 * h] = t.x; v[4 * h + 1] = t.y; v[4 * h + 2] = t.z; v[4 * h + 3] = t.w;
    }
}
template <int N> __device__ __forceinline__ void load_frames(const bf16_t* __restrict__ p, float (&v)[N]) {
    static_assert(N == 4 || N == 8, "4 or 8 frames per access");
    if constexpr (N == 8) {
        const u4v t = *reinterpret_cast<const u4v*>(p);
        v[0] = bf16_lo(t.x); v[1] = bf16_hi(t.x); v[2] = bf16_lo(t.y); v[3] = bf16_hi(t.y);
        v[4] = bf16_lo(t.z); v[5] = bf16_hi(t.z); v[6] = bf16_lo(t.w); v[7] = bf16_hi(t.w);
    } else {
        const u2v t = *reinterpret_cast<const u2v*>(p);
        v[0] = bf16_lo(t.x); v[1] = bf16_hi(t.x); v[2] = bf16_lo(t.y); v[3] = bf16_hi(t.y);
    }
}
template <int N, typename T> __device__ __forceinline__ void zero_frames(float (&v)[N]) {
#pragma unroll
    for (int r = 0; r < N; ++r) v[r] = 0.f;
}

// NT: non-temporal (streaming) store -- the output of a node is read next by another kernel, not by this one
template <int N, bool NT> __device__ __forceinline__ void store_frames(float* __restrict__ p, const float (&v)[N]) {
    if constexpr (N == 2) {
        const f2v t = {v[0], v[1]};
        if (NT) __builtin_nontemporal_store(t, reinterpret_cast<f2v*>(p));
        else *reinterpret_cast<f2v*>(p) = t;
        return;
    }
#pragma unroll
    for (int h = 0; h < N / 4; ++h) {
        const f4v t = {v[4 * h], v[4 * h + 1], v[4 * h + 2], v[4 * h + 3]};
        if (NT) __builtin_nontemporal_store(t, reinterpret_cast<f4v*>(p + 4 * h));
        else *reinterpret_cast<f4v*>(p + 4 * h) = t;
    }
}
template <int N, bool NT> __device__ __forceinline__ void store_frames(bf16_t* __restrict__ p, const float (&v)[N]) {
    if constexpr (N == 8) {
        const u4v t = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
        if (NT) __builtin_nontemporal_store(t, reinterpret_cast<u4v*>(p));
        else *reinterpret_cast<u4v*>(p) = t;
    } else {
        const u2v t = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
        if (NT) __builtin_nontemporal_store(t, reinterpret_cast<u2v*>(p));
        else *reinterpret_cast<u2v*>(p) = t;
    }
}

}  // namespace nbasr
