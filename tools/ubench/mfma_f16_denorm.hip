// Does v_mfma_f32_16x16x32_f16 honour fp16 subnormal INPUTS?  (decides whether the two-way fp16 split of the dense conv
// can carry tiny operands exactly).  Build: hipcc --offload-arch=gfx950 -O2 mfma_f16_denorm.hip -o mfma_f16_denorm
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
__global__ void k(float a_val, float b_val, float* out)
{
    halfx8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)0.f; b[i] = (_Float16)0.f; }
    a[0] = (_Float16)a_val;          // k = 8 * (lane >> 4): only lanes 0..15 carry k = 0
    b[0] = (_Float16)b_val;
    if (threadIdx.x >= 16) { a[0] = (_Float16)0.f; b[0] = (_Float16)0.f; }
    floatx4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = c[0]; out[1] = (float)a[0]; }
}
int main()
{
    float* d; hipMalloc(&d, 8);
    const float vals[] = {9.5367431640625e-07f /* 2^-20: subnormal */, 5.9604644775390625e-08f /* 2^-24: smallest */, 6.103515625e-05f /* min normal */};
    for (float v : vals) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, v, 1024.f, d);
        float h[2]; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
        printf("a=%g (as fp16 %g) * 1024 -> mfma %g (exact %g)\n", v, h[1], h[0], v * 1024.f);
    }
    return 0;
}
