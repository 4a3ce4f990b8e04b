// Microbenchmark (dev aid, not product code): what does the CG direction step's memory shape -- read r, p, x; write p, x, in place --
// reach on vectors far beyond the Infinity Cache (n = 64 M fp64: 512 MB each), as a function of the launch shape and the cache
// policy?  Shapes: the product's (one workgroup per reduction chunk of 32768 elements, 1954 workgroups, 64 sixteen-byte steps per
// thread) against short workgroups on a large grid (1, 2, 4, 8 steps per thread).  Plus copy (1R 1W) and triad (2R 1W) ceilings.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/axpy_probe tools/ubench/axpy_probe.hip ; run: tools/ubench/axpy_probe [n]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); exit(1); } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));

template <bool NT> __device__ __forceinline__ d2 ld(const double *p, long i) {
    return NT ? __builtin_nontemporal_load((const d2 *)(p + i)) : *(const d2 *)(p + i);
}
template <bool NT> __device__ __forceinline__ void st(double *p, long i, d2 v) {
    if (NT) __builtin_nontemporal_store(v, (d2 *)(p + i)); else *(d2 *)(p + i) = v;
}
// MODE 0: copy y = x; 1: triad y = a + s b; 2: direction (x += alpha p; p = r + beta p)
template <int MODE, int STEPS, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void k(const double *__restrict__ r, double *__restrict__ p, double *__restrict__ x, long n, double alpha, double beta) {
    const long base = (long)blockIdx.x * 512 * STEPS + 2 * threadIdx.x;
#pragma unroll 4
    for (int j = 0; j < STEPS; ++j) {
        const long i = base + 512L * j;
        if (i + 1 < n) {
            if (MODE == 0) { st<NTS>(x, i, ld<NTL>(r, i)); }
            else if (MODE == 1) { d2 a = ld<NTL>(r, i), b = ld<NTL>(p, i); st<NTS>(x, i, a + beta * b); }
            else {
                d2 rv = ld<NTL>(r, i), pv = ld<NTL>(p, i), xv = ld<NTL>(x, i);
                st<NTS>(x, i, xv + alpha * pv);
                st<false>(p, i, rv + beta * pv);
            }
        }
    }
}
template <int MODE, int STEPS, bool NTL, bool NTS>
static void run(const char *name, double *r, double *p, double *x, long n, double bytes) {
    const long per = 512L * STEPS;
    const int grid = (int)((n + per - 1) / per);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int w = 0; w < 3; ++w) k<MODE, STEPS, NTL, NTS><<<grid, 256>>>(r, p, x, n, 1e-9, 0.5);
    CK(hipEventRecord(a));
    const int reps = 20;
    for (int w = 0; w < reps; ++w) k<MODE, STEPS, NTL, NTS><<<grid, 256>>>(r, p, x, n, 1e-9, 0.5);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double us = ms / reps * 1e3;
    printf("%-34s steps %3d grid %7d ntl %d nts %d : %8.1f us  %6.2f TB/s\n", name, STEPS, grid, (int)NTL, (int)NTS, us, bytes / us / 1e6);
}
int main(int argc, char **argv) {
    const long n = argc > 1 ? atol(argv[1]) : 64000000L;
    double *r, *p, *x;
    CK(hipMalloc(&r, n * 8)); CK(hipMalloc(&p, n * 8)); CK(hipMalloc(&x, n * 8));
    CK(hipMemset(r, 0, n * 8)); CK(hipMemset(p, 0, n * 8)); CK(hipMemset(x, 0, n * 8));
    const double B = (double)n * 8;
    printf("n = %ld (%.0f MB per vector)\n", n, B / 1e6);
#define ALL(MODE, name, bytes)                                     \
    run<MODE, 64, false, false>(name, r, p, x, n, bytes);          \
    run<MODE, 64, true, true>(name, r, p, x, n, bytes);            \
    run<MODE, 8, false, false>(name, r, p, x, n, bytes);           \
    run<MODE, 8, true, true>(name, r, p, x, n, bytes);             \
    run<MODE, 4, true, true>(name, r, p, x, n, bytes);             \
    run<MODE, 2, false, false>(name, r, p, x, n, bytes);           \
    run<MODE, 2, true, true>(name, r, p, x, n, bytes);             \
    run<MODE, 2, true, false>(name, r, p, x, n, bytes);            \
    run<MODE, 2, false, true>(name, r, p, x, n, bytes);            \
    run<MODE, 1, true, true>(name, r, p, x, n, bytes);
    if (argc > 2 && argv[2][0] == 'r') {  // re-allocation sweep: does the time of the SAME kernel depend on where the vectors land?
        for (int rep = 0; rep < 6; ++rep) {
            double *a[3], *dummy = nullptr;
            if (rep % 2) CK(hipMalloc(&dummy, (size_t)(rep * 37 + 11) << 20));  // shift what the next allocations get
            for (int k = 0; k < 3; ++k) { CK(hipMalloc(&a[k], n * 8)); CK(hipMemset(a[k], 0, n * 8)); }
            char name[64];
            snprintf(name, sizeof(name), "direction, separate allocs #%d", rep);
            run<2, 64, true, true>(name, a[0], a[1], a[2], n, 5 * B);
            printf("   r %p p %p x %p\n", (void *)a[0], (void *)a[1], (void *)a[2]);
            for (int k = 0; k < 3; ++k) CK(hipFree(a[k]));
            double *buf;
            CK(hipMalloc(&buf, 3 * n * 8)); CK(hipMemset(buf, 0, 3 * n * 8));
            snprintf(name, sizeof(name), "direction, one allocation #%d", rep);
            run<2, 64, true, true>(name, buf, buf + n, buf + 2 * n, n, 5 * B);
            printf("   buf %p\n", (void *)buf);
            CK(hipFree(buf));
            if (dummy) CK(hipFree(dummy));
        }
        return 0;
    }
    if (argc > 2) {  // skew sweep: the three vectors in ONE allocation, p and x displaced by k and 2k times `skew` bytes from an n-vector stride
        double *buf;
        const long pad = 64L << 20;
        CK(hipMalloc(&buf, 3 * n * 8 + 3 * pad));
        CK(hipMemset(buf, 0, 3 * n * 8 + 3 * pad));
        const long skews[] = {0, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 65536, 1L << 20, (1L << 20) + 4096, 3L << 20};
        for (long sk : skews) {
            double *rr = buf, *pp = buf + n + sk / 8, *xx = buf + 2 * n + 2 * sk / 8;
            char name[64];
            snprintf(name, sizeof(name), "direction, skew %ld B", sk);
            run<2, 64, true, true>(name, rr, pp, xx, n, 5 * B);
            run<2, 4, true, true>(name, rr, pp, xx, n, 5 * B);
        }
        return 0;
    }
    ALL(0, "copy 1R 1W", 2 * B)
    ALL(1, "triad 2R 1W", 3 * B)
    ALL(2, "direction 3R 2W (in place)", 5 * B)
    return 0;
}
