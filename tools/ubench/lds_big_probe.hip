// Probe: (1) can a kernel use more than 64 KB of dynamic LDS on gfx950 (with / without hipFuncSetAttribute), and how many such
// blocks share a CU; (2) the rate of s_memtime against wall time for a lightly loaded chip (40 blocks).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__global__ __launch_bounds__(256, 2) void big_lds(unsigned long long *out, int words, int spin) {
    extern __shared__ double sm[];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = threadIdx.x; i < words; i += blockDim.x) sm[i] = (double)i;
    __syncthreads();
    double acc = 0.0;
    for (int r = 0; r < spin; ++r)
        for (int i = threadIdx.x; i < words; i += blockDim.x) acc += sm[(i + r) % words];
    unsigned xcc, hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) {
        out[blockIdx.x * 4 + 0] = t1 - t0;
        out[blockIdx.x * 4 + 1] = xcc & 15u;
        out[blockIdx.x * 4 + 2] = hwid;
        out[blockIdx.x * 4 + 3] = (unsigned long long)acc;
    }
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("sharedMemPerBlock %zu  maxSharedMemoryPerMultiProcessor %zu  sharedMemPerBlockOptin %zu  clockRate %d kHz  CUs %d\n",
           p.sharedMemPerBlock, p.maxSharedMemoryPerMultiProcessor, p.sharedMemPerBlockOptin, p.clockRate, p.multiProcessorCount);
    unsigned long long *out;
    hipMalloc(&out, 4096 * 4 * sizeof(unsigned long long));
    unsigned long long *h = (unsigned long long *)malloc(4096 * 4 * sizeof(unsigned long long));
    for (int kb : {60, 72, 80, 96, 150}) {
        const size_t bytes = (size_t)kb * 1024;
        hipError_t e0 = hipFuncSetAttribute((const void *)big_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        hipEvent_t a, b;
        hipEventCreate(&a);
        hipEventCreate(&b);
        const int blocks = 320;   // the cycle kernel's grid at g = 5: 40 working blocks of 320
        hipMemset(out, 0, 4096 * 4 * sizeof(unsigned long long));
        hipEventRecord(a);
        big_lds<<<blocks, 256, bytes>>>(out, (int)(bytes / 8), 200);
        hipError_t e1 = hipGetLastError();
        hipEventRecord(b);
        hipError_t e2 = hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        hipMemcpy(h, out, blocks * 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        unsigned long long tmax = 0;
        int per_xcc[16] = {0};
        for (int i = 0; i < blocks; ++i) {
            if (h[i * 4] > tmax) tmax = h[i * 4];
            per_xcc[h[i * 4 + 1] & 15]++;
        }
        int rr = 1;
        for (int i = 0; i < blocks; ++i) rr &= ((int)(h[i * 4 + 1]) == (i & 7));
        printf("LDS %3d KB: setattr %s launch %s sync %s  kernel %.3f ms  max block clocks %llu -> %.0f MHz if one wave of blocks  "
               "xcc(b)==b%%8 %s  blocks/xcc %d %d %d %d %d %d %d %d\n",
               kb, hipGetErrorName(e0), hipGetErrorName(e1), hipGetErrorName(e2), ms, tmax, tmax / (ms * 1e3), rr ? "yes" : "NO",
               per_xcc[0], per_xcc[1], per_xcc[2], per_xcc[3], per_xcc[4], per_xcc[5], per_xcc[6], per_xcc[7]);
    }
    return 0;
}
