// Micro-benchmark: what does a bare v_mfma_f32_32x32x2_f32 loop sustain on this chip, alone and with the
// ds_read_b32 traffic of the conv GEMM's inner loop?  (diagnostic, not part of the library)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(float* out, int iters) {
    __shared__ float lds[64 * 129];
    for (int i = threadIdx.x; i < 64 * 129; i += 256) lds[i] = 0.001f * i;
    __syncthreads();
    floatx16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a0 = threadIdx.x * 0.5f, a1 = a0 + 1.f, b0 = 0.25f, b1 = 0.75f;
    const float* base = lds + (threadIdx.x & 63);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int kk = 0; kk < 32; ++kk) {
            if (MODE == 1) { a0 = base[kk * 129]; a1 = base[kk * 129 + 32]; b0 = base[kk * 129 + 64]; b1 = base[kk * 129 + 65]; }
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[3], 0, 0, 0);
        }
        if (MODE == 2) __syncthreads();
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int blocks) {
    float* out; hipMalloc(&out, blocks * 256 * sizeof(float));
    const int iters = 400;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * 4 /*waves*/ * iters * 32 * 4 * 4096.0;
    printf("%-40s blocks=%4d  %8.3f ms  %7.1f TFLOP/s\n", name, blocks, ms, flops / ms / 1e9);
    hipFree(out);
}
int main() {
    run<0>("bare MFMA, 1 WG/CU", 256);
    run<0>("bare MFMA, 2 WG/CU", 512);
    run<1>("MFMA + 4 ds_read_b32 / 4 MFMA, 2 WG/CU", 512);
    run<2>("bare MFMA + barrier per 128 MFMA, 2 WG/CU", 512);
    run<0>("bare MFMA, 4 WG/CU (oversubscribed)", 1024);
    return 0;
}
