// Issue rate of the packed-bf16 dot instructions of gfx950 against v_pk_fma_f32 / v_fma_f32: is two bf16 MACs per lane and
// instruction (v_dot2c_f32_bf16 / v_dot2_f32_bf16: D = a.lo * b.lo + a.hi * b.hi + C, fp32 accumulate) faster per MAC than the packed
// fp32 FMA the bf16 node kernel runs on?  (bf16 x bf16 products are exact in fp32; the weights of a bf16 model ARE bf16 values.)
// build: hipcc --offload-arch=gfx950 -O3 -o dot2_rate.bin dot2_rate.hip ; run: ./dot2_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void rate_kernel(float* out, const float* wsrc, int iters)
{
    float acc[32];
    for (int i = 0; i < 32; ++i) acc[i] = float(i) + threadIdx.x * 1e-3f;
    const unsigned xb = 0x3f803f00u + (threadIdx.x & 15);         // two bf16 values (1.0, 0.5 + eps)
    const float ws = wsrc[blockIdx.x & 1];                         // wave-uniform -> SGPR
    const unsigned wb = __float_as_uint(ws) | 0x3f000000u;         // packed bf16 pair in an SGPR
    f2 x2 = f2{1.0f + threadIdx.x * 1e-6f, 0.5f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                if (MODE == 0)        // dot2c: 2 MACs, accumulator is the destination (VOP2)
                    asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(acc[i]) : "s"(wb), "v"(xb));
                else if (MODE == 1)   // dot2 (VOP3P): 2 MACs
                    asm volatile("v_dot2_f32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "s"(wb), "v"(xb));
                else if (MODE == 2)   // one fp32 FMA with a scalar weight (1 MAC)
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "s"(ws), "v"(x2.x));
                else if ((i & 1) == 0) {   // packed fp32 FMA on a register pair (2 MACs per instruction, 16 instructions)
                    f2 a = f2{acc[i], acc[i + 1]};
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(a) : "s"(f2{ws, ws}), "v"(x2));
                    acc[i] = a.x; acc[i + 1] = a.y;
                }
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 32; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, float* out, float* w, double macs_per_inner)
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
    const double macs = double(blocks) * 256 * iters * 4 * macs_per_inner;
    printf("%-52s %8.3f ms  %7.1f TMAC/s  (%.1f TFLOP/s)\n", name, ms, macs / ms / 1e9, 2 * macs / ms / 1e9);
}

int main()
{
    float *out, *w;
    hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    hipMalloc(&w, 8);
    const float hw[2] = {0.999f, 1.001f};
    hipMemcpy(w, hw, 8, hipMemcpyHostToDevice);
    run<0>("v_dot2c_f32_bf16 (2 MACs / lane / instr)", out, w, 64);
    run<1>("v_dot2_f32_bf16  (2 MACs / lane / instr)", out, w, 64);
    run<2>("v_fma_f32        (1 MAC  / lane / instr)", out, w, 32);
    run<3>("v_pk_fma_f32     (2 MACs / lane / instr)", out, w, 32);
    return 0;
}
