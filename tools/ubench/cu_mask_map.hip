// Which (XCC, SE, CU) does bit i of a hipExtStreamCreateWithCUMask mask select?  Each launch runs 2048 small workgroups on
// a stream masked to ONE bit and records the hardware ids they landed on.
// Build: hipcc --offload-arch=gfx950 -O2 cu_mask_map.hip -o cu_mask_map.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <set>
#include <vector>
__global__ void where(unsigned* out)
{
    if (threadIdx.x == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        out[blockIdx.x * 2] = hw;
        out[blockIdx.x * 2 + 1] = xcc;
    }
    for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(1);
}
int main()
{
    const int n = 2048;
    unsigned* d; hipMalloc(&d, n * 8);
    std::vector<unsigned> h(n * 2);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("multiProcessorCount %d\n", p.multiProcessorCount);
    for (int bit = 0; bit < 256; bit += (bit < 40 ? 1 : 37)) {
        unsigned mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        mask[bit / 32] = 1u << (bit % 32);
        hipStream_t s;
        if (hipExtStreamCreateWithCUMask(&s, 8, mask) != hipSuccess) { printf("bit %d: stream creation failed\n", bit); continue; }
        hipLaunchKernelGGL(where, dim3(n), dim3(64), 0, s, d);
        hipStreamSynchronize(s);
        hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost);
        std::set<unsigned> ids;
        for (int i = 0; i < n; ++i) {
            const unsigned hw = h[i * 2], xcc = h[i * 2 + 1] & 0xf;
            const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 0x1, se = (hw >> 13) & 0x7;
            ids.insert((xcc << 12) | (se << 8) | (sh << 4) | cu);
        }
        printf("bit %3d ->", bit);
        for (unsigned id : ids) printf(" xcc%u/se%u/sh%u/cu%u", id >> 12, (id >> 8) & 0xf, (id >> 4) & 0xf, id & 0xf);
        printf("\n");
        hipStreamDestroy(s);
    }
    return 0;
}
