// Exploration for the next round (not part of the library): the pipelined, output-split fp32 node kernel of
// nb_asr_amd/csrc/grouped_conv_osplit.hip reduced to its plain flavour (no LayerNorm on load, no skips, no statistics), with one
// switch: WPF = the scalar weight loads of channel ci + 1 are issued BEFORE the FMAs of channel ci, into a second register set
// (2 x CO x K SGPRs; fits only with the output split: 60 at CG = 12), the way the window loads already are.  DESIGN.md section 7.1 names
// the weight loads -- waited for in full before a channel's first FMA -- as the next serial piece of a wave's loop.
//   hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -shared -I nb_asr_amd/csrc -I include -x hip tools/ubench/x1/gc_wprefetch.hip -o tools/ubench/x1/libgc_wpf.so
#include <hip/hip_runtime.h>
#include "common.h"

namespace x1 {
using nbasr::pad_left;
using nbasr::relu_clamp;

typedef float gc_f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float4 row_chunk(const float* row, int row_bytes, int byte_offset)
{
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(row), 0, row_bytes, 0x00020000);
    const gc_f4 f = __builtin_bit_cast(gc_f4, __builtin_amdgcn_raw_buffer_load_b128(rs, byte_offset, 0, 0));
    return make_float4(f[0], f[1], f[2], f[3]);
}

template <int CG, int K, int D, bool WPF>
__global__ __launch_bounds__(256) void node_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                   float* __restrict__ y, int channels, int frames, int ld, int groups)
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
static int launch(int wpf, const float* x, const float* w, const float* bias, float* y, int batch, int channels, int frames, int ld, hipStream_t s)
{
    const int groups = channels / CG, nq = ld >> 2;
    const dim3 grid((nq + 63) / 64, (groups + 1) / 2, batch);
    if (wpf) hipLaunchKernelGGL((node_kernel<CG, 5, 1, true>), grid, dim3(256), 0, s, x, w, bias, y, channels, frames, ld, groups);
    else     hipLaunchKernelGGL((node_kernel<CG, 5, 1, false>), grid, dim3(256), 0, s, x, w, bias, y, channels, frames, ld, groups);
    return static_cast<int>(hipGetLastError());
}
}  // namespace x1

extern "C" int x1_node(int wpf, const float* x, const float* w, const float* bias, float* y, int batch, int channels, int frames, int ld,
                       void* stream)
{
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (channels / 100) {
        case 6:  return x1::launch<6>(wpf, x, w, bias, y, batch, channels, frames, ld, s);
        case 8:  return x1::launch<8>(wpf, x, w, bias, y, batch, channels, frames, ld, s);
        case 10: return x1::launch<10>(wpf, x, w, bias, y, batch, channels, frames, ld, s);
        case 12: return x1::launch<12>(wpf, x, w, bias, y, batch, channels, frames, ld, s);
    }
    return -1;
}
