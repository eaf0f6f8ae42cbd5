// v_pk_fma_f32 issue rate on gfx950 by operand form: does a scalar (SGPR) weight operand, as the grouped convolution uses it,
// cost issue slots compared with an all-VGPR form?  And what does plain v_fma_f32 reach?
// build: hipcc --offload-arch=gfx950 -O3 -o pk_fma_rate.bin pk_fma_rate.hip ; run: ./pk_fma_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void rate_kernel(float* out, const float* wsrc, int iters)
{
    f2 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f2{float(i), float(threadIdx.x)};
    f2 x = f2{1.0f + threadIdx.x * 1e-6f, 0.5f};
    const float ws = wsrc[blockIdx.x & 1];                 // wave-uniform -> SGPR
    f2 wv = f2{wsrc[threadIdx.x & 1], 0.f};                // per-lane -> VGPR
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (MODE == 0)        // SGPR weight, both halves use its low dword (the grouped conv's form)
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[i]) : "s"(f2{ws, ws}), "v"(x));
                else if (MODE == 1)   // VGPR weight, low dword for both halves
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[i]) : "v"(wv), "v"(x));
                else if (MODE == 2)   // all VGPR, plain packed
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(wv), "v"(x));
                else {                // two scalar-form FMAs (same flops as one packed)
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i].x) : "s"(ws), "v"(x.x));
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i].y) : "s"(ws), "v"(x.y));
                }
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i].x + acc[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, float* out, float* w)
{
    const int blocks = 256 * 8, iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, out, w, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, out, w, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = double(blocks) * 256 * iters * 64 * 4;      // 64 packed FMAs = 256 flops per thread and iteration
    printf("%-44s %8.3f ms  %7.1f TFLOP/s\n", name, ms, flops / ms / 1e9);
}

int main()
{
    float *out, *w;
    hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    hipMalloc(&w, 8);
    float hw[2] = {1.0000001f, 0.9999999f};
    hipMemcpy(w, hw, 8, hipMemcpyHostToDevice);
    run<0>("v_pk_fma_f32 sgpr weight (op_sel_hi 0,1,1)", out, w);
    run<1>("v_pk_fma_f32 vgpr weight (op_sel_hi 0,1,1)", out, w);
    run<2>("v_pk_fma_f32 all vgpr, plain", out, w);
    run<3>("2 x v_fma_f32 sgpr weight", out, w);
    return 0;
}
