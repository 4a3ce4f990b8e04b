// Dependent-chain latencies on gfx950 (one wavefront, lane 0 meaningful): what one "round trip" of the small-system GMRES
// kernel's hand-offs and gathers costs.  Each chain is 2000 dependent operations on the same few cache lines.
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ void chain(unsigned long long *buf, unsigned long long *out, int mode) {
    __shared__ unsigned long long lds[64];
    lds[threadIdx.x & 63] = 0;
    __syncthreads();
    unsigned long long v = 0;
    const int N = 2000;
    // warm
    for (int i = 0; i < 16; ++i) v += __hip_atomic_load(buf + (v & 7), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < N; ++i) {
        unsigned long long *p = buf + (v & 7);
        unsigned long long x;
        if (mode == 0) x = *(volatile unsigned long long *)p;                                                   // (volatile: sc0 sc1)
        else if (mode == 1) x = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);           // plain
        else if (mode == 2) x = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);               // sc1
        else if (mode == 3) x = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);              // sc0 sc1
        else if (mode == 4) x = __hip_atomic_fetch_add(p, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // returning atomic
        else if (mode == 5) x = lds[v & 7];                                                                     // LDS
        else if (mode == 6) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);         // plain store + wait + sc1 load
                              asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                              x = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0; }
        else { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);                            // sc1 store + wait + sc1 load
               asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
               x = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0; }
        v += x;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) {
        out[0] = (t1 - t0) / N;
        out[1] = v;
    }
}

// arithmetic chains: dependent fp64 fma / division / sqrt
__global__ void arith(double *out, int mode) {
    double v = 1.0 + threadIdx.x * 1e-9, w = 1.000001;
    const int N = 2000;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < N; ++i) {
        if (mode == 0) v = fma(v, w, 1e-9);
        else if (mode == 1) v = v / w;
        else v = sqrt(v) + 1.0;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) {
        out[0] = (double)((t1 - t0) / N);
        out[1] = v;
    }
}

int main() {
    unsigned long long *buf, *out, h[2];
    double *dout, hd[2];
    hipMalloc(&buf, 4096);
    hipMemset(buf, 0, 4096);
    hipMalloc(&out, 64);
    hipMalloc(&dout, 64);
    const char *names[] = {"volatile load (sc0 sc1)", "plain load (L1 hit)", "sc1 load (agent scope)", "sc0 sc1 load (system scope)",
                           "returning atomic add, agent", "LDS load", "plain store + vmcnt(0) + sc1 load", "sc1 store + vmcnt(0) + sc1 load"};
    for (int m = 0; m < 8; ++m) {
        chain<<<1, 64>>>(buf, out, m);
        hipDeviceSynchronize();
        hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
        printf("%-40s %6llu clocks per dependent operation\n", names[m], h[0]);
    }
    const char *an[] = {"fp64 fma", "fp64 division", "fp64 sqrt + add"};
    for (int m = 0; m < 3; ++m) {
        arith<<<1, 64>>>(dout, m);
        hipDeviceSynchronize();
        hipMemcpy(hd, dout, 16, hipMemcpyDeviceToHost);
        printf("%-40s %6.0f clocks per dependent operation\n", an[m], hd[0]);
    }
    return 0;
}
