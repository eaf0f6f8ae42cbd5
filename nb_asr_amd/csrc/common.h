// Shared helpers for the gfx950 kernels of libnbasr_hip.so (internal; the public ABI is include/nbasr.h).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "nbasr.h"

namespace nbasr {

// thread-local error text behind nbasr_last_error()
void set_error(const char* fmt, ...);
void clear_error();

inline hipStream_t as_stream(nbasr_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

inline int launch_status(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return static_cast<int>(e);
    }
    return NBASR_OK;
}

// host + device copy of the PadConvRelu padding rule (reference ops.py:12-17, context = 4)
__host__ __device__ constexpr int pad_left(int kernel, int dilation, int stride) {
    return (4 / stride >= kernel * dilation - stride) ? 0 : (kernel - 1) * dilation - 4 / stride;
}
__host__ __device__ constexpr int pad_right(int kernel, int dilation, int stride) {
    return (4 / stride >= kernel * dilation - stride) ? kernel * dilation - stride : 4 / stride;
}

constexpr float kClamp = 20.0f;   // reference ops.py:28

// device-side view of nbasr_deferred_ln (stats == nullptr: no pending LayerNorm)
struct LnRef {
    const float* stats; const float* gamma; const float* beta;
};
inline LnRef ln_ref(const nbasr_deferred_ln* ln, bool wanted) {
    return (ln && wanted) ? LnRef{ln->stats, ln->gamma, ln->beta} : LnRef{nullptr, nullptr, nullptr};
}
// xn = (x - mean) * rstd * gamma + beta, exactly 0 where rstd == 0 (pitch columns / frames outside the utterance)
__device__ __forceinline__ float ln_apply(float x, float mean, float rstd, float gamma, float beta) {
    return rstd != 0.f ? __builtin_fmaf((x - mean) * rstd, gamma, beta) : 0.f;
}

// min(relu(v), 20): v_med3_f32 (fminf(fmaxf(v, 0), 20) costs an extra canonicalising v_max).  v_med3 alone maps NaN to 0
// (it returns min3 of its operands when one is NaN), which would turn a diverged checkpoint or a corrupt input into
// finite-looking logits; the reference's relu / clamp_max_ (ops.py:27-28) propagate NaN.  fma(v, 0, med3) restores that for
// one more (packable) instruction: v * 0 is +-0 for finite v -- the sum is then exactly med3 -- and NaN for NaN.  It is NaN
// for +-Inf too, where the reference gives 20 / 0: an overflowed pre-activation comes out LOUDER here, never quieter.
// NBASR_NAN_QUIET=1 (compile time) restores the bare v_med3 for A/B timing.
#ifndef NBASR_NAN_QUIET
#define NBASR_NAN_QUIET 0
#endif
__device__ __forceinline__ float relu_clamp(float v) {
    const float m = __builtin_amdgcn_fmed3f(v, 0.0f, kClamp);
    return NBASR_NAN_QUIET ? m : __builtin_fmaf(v, 0.0f, m);
}
// |v| if it is finite, else 0: range statistics (max|x| per utterance) must not be hijacked by an Inf / NaN sample --
// the scaled fp16 GEMM would otherwise flush every finite sample of that utterance to zero
__device__ __forceinline__ float finite_abs(float v) { const float a = fabsf(v); return a <= 3.4028234664e38f ? a : 0.f; }

#define NBASR_REQUIRE(cond, code, ...)          \
    do {                                        \
        if (!(cond)) {                          \
            ::nbasr::set_error(__VA_ARGS__);    \
            return (code);                      \
        }                                       \
    } while (0)

}  // namespace nbasr
