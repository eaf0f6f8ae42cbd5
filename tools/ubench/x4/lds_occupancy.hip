// How many workgroups of a given dynamic-LDS size does a gfx950 CU hold?  (the allocation granularity of LDS decides whether a fifth
// 32 128-byte workgroup or a thirteenth 12 288-byte one fits the 160 KiB)
// build + run: hipcc --offload-arch=gfx950 -O2 -o lds_occupancy.bin lds_occupancy.hip && ./lds_occupancy.bin
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void probe(float* out)
{
    extern __shared__ float lds[];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    out[blockIdx.x * blockDim.x + threadIdx.x] = lds[(threadIdx.x + 1) % blockDim.x];
}

int main()
{
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int sizes[] = {12288, 12544, 12672, 12800, 13056, 16384, 20160, 21120, 24096, 32000, 32128, 32256, 32384, 32768, 33280, 40960, 49152, 53248, 54784, 65536, 81920};
    for (int threads : {64, 128, 256, 512}) {
        for (int bytes : sizes) {
            int n = 0;
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, probe, threads, bytes);
            printf("threads %4d  lds %6d B  -> %2d workgroups per CU (%6d B in all)\n", threads, bytes, n, n * bytes);
        }
    }
    return 0;
}
