// hipcc (ROCm 7.2, gfx950): __builtin_bit_cast(unsigned, v[e]) where v is an ext_vector_type(4) float and v[e] an ELEMENT reads element 0 for every e
// (-DV1: the four tests fold into nothing); __float_as_uint(v[e]) is right (8 v_xnor).  hipcc -O3 --offload-arch=gfx950 --cuda-device-only -S.
// The same quirk is behind "a raw_buffer_load_b128 result assigned to an unsigned ext-vector degrades to one dword": its elements were read this way.
#include <hip/hip_runtime.h>
typedef float floatx4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* p, unsigned* out, unsigned want) {
    floatx4 hr[2];
    for (int c = 0; c < 2; ++c) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(hr[c]) : "v"(p + (threadIdx.x + c * 64) * 4) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(hr[0]), "+v"(hr[1]) :: "memory");
    unsigned t0 = 0x40000000u, t1 = 0x40000000u;
#ifdef V1
    for (int e = 0; e < 4; ++e) { t0 &= __builtin_bit_cast(unsigned, hr[0][e]) ^ ~want; t1 &= __builtin_bit_cast(unsigned, hr[1][e]) ^ ~want; }
#else
    for (int e = 0; e < 4; ++e) { t0 &= __float_as_uint(hr[0][e]) ^ ~want; t1 &= __float_as_uint(hr[1][e]) ^ ~want; }
#endif
    out[threadIdx.x] = t0 + 3 * t1;
}
