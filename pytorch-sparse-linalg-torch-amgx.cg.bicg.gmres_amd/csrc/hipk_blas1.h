// hipk_blas1.h -- chunked element access helpers + BLAS-1 style launchers.
#pragma once
#include "hipk_common.h"

#ifdef __HIPCC__
// 16-byte vector access to VEC consecutive elements starting at element i (i % VEC == 0,
// base pointer 16-byte aligned); falls back to scalars for the ragged tail (nv < VEC).
template <typename T>
__device__ __forceinline__ void hipk_ld(const T *__restrict__ p, int64_t i, int nv,
                                        T (&out)[hipk_vec<T>::VEC]) {
    constexpr int VEC = hipk_vec<T>::VEC;
    if (nv == VEC) {
        const typename hipk_vec<T>::type v = *(const typename hipk_vec<T>::type *)(p + i);
        const T *vp = (const T *)&v;
#pragma unroll
        for (int k = 0; k < VEC; ++k) out[k] = vp[k];
    } else {
#pragma unroll
        for (int k = 0; k < VEC; ++k) out[k] = (k < nv) ? p[i + k] : (T)0;
    }
}

template <typename T>
__device__ __forceinline__ void hipk_st(T *__restrict__ p, int64_t i, int nv,
                                        const T (&in)[hipk_vec<T>::VEC]) {
    constexpr int VEC = hipk_vec<T>::VEC;
    if (nv == VEC) {
        typename hipk_vec<T>::type v;
        T *vp = (T *)&v;
#pragma unroll
        for (int k = 0; k < VEC; ++k) vp[k] = in[k];
        *(typename hipk_vec<T>::type *)(p + i) = v;
    } else {
        for (int k = 0; k < nv; ++k) p[i + k] = in[k];
    }
}

// Non-temporal variants: streams that are dead after this access should not occupy the Infinity Cache.
template <typename T>
__device__ __forceinline__ void hipk_ld_nt_vec(const T *__restrict__ p, int64_t i, int nv, T (&out)[hipk_vec<T>::VEC]) {
    constexpr int VEC = hipk_vec<T>::VEC;
    if (nv == VEC) {
        typedef T native_t __attribute__((ext_vector_type(hipk_vec<T>::VEC)));
        const native_t v = __builtin_nontemporal_load((const native_t *)(p + i));
        const T *vp = (const T *)&v;
#pragma unroll
        for (int k = 0; k < VEC; ++k) out[k] = vp[k];
    } else {
#pragma unroll
        for (int k = 0; k < VEC; ++k) out[k] = (k < nv) ? p[i + k] : (T)0;
    }
}

template <typename T>
__device__ __forceinline__ void hipk_st_nt_vec(T *__restrict__ p, int64_t i, int nv, const T (&in)[hipk_vec<T>::VEC]) {
    constexpr int VEC = hipk_vec<T>::VEC;
    if (nv == VEC) {
        typedef T native_t __attribute__((ext_vector_type(hipk_vec<T>::VEC)));
        native_t v;
#pragma unroll
        for (int k = 0; k < VEC; ++k) v[k] = in[k];
        __builtin_nontemporal_store(v, (native_t *)(p + i));
    } else {
        for (int k = 0; k < nv; ++k) p[i + k] = in[k];
    }
}

// Iterate the calling thread's elements of chunk c in reduction-spec order.
// f(int64_t i, int nv): i = first element, nv = valid elements (1..VEC).
template <typename T, int UNROLL = 4, typename F>
__device__ __forceinline__ void hipk_chunk_loop(int64_t n, int ch, int c, F f) {
    constexpr int VEC = hipk_vec<T>::VEC;
    const int64_t base = (int64_t)c * ch;
    const int64_t end = (base + ch < n) ? base + ch : n;
#pragma unroll UNROLL
    for (int64_t i = base + (int64_t)VEC * threadIdx.x; i < end; i += (int64_t)VEC * HIPK_THREADS) {
        const int nv = (end - i < VEC) ? (int)(end - i) : VEC;
        f(i, nv);
    }
}

// Chunk loop whose first HIPK_BASE_CHUNK elements of NV operands are requested EARLY: the per-iteration vector
// kernels run as one wave of workgroups (a chunk each), so everything a workgroup does before its first vector
// load -- stop word, partial sums, the fold's barriers -- would be exposed in full.  Usage:
//     hipk_pre<T, NV> pre;  pre.issue(n, ch, c, {a, b});      // loads in flight
//     ... stop test, fold of the partials ...
//     pre.run([&](int64_t i, int nv, T (&v)[NV][VEC]) { ... });   // same element order as hipk_chunk_loop
// Larger chunks continue with ordinary loads.  NV = 2 keeps the kernels at 8 workgroups per CU.
// NT: the operands are STREAMS (the iteration's working set is far larger than the 256 MiB Infinity Cache, N >> 8 M rows):
// non-temporal loads, so that they do not displace each other's lines on their way through.
template <typename T, int NV, bool NT = false, int NB = 0>
struct hipk_pre {
    static constexpr int VEC = hipk_vec<T>::VEC;
    static constexpr int N0 = HIPK_BASE_CHUNK / (VEC * HIPK_THREADS);  // steps of the smallest chunk
    // register-resident steps per thread: the smallest chunk's, all requested up front; with the streaming policy optionally
    // fewer (NB: more operands in the same registers), the rest then arrives in batches of as many steps (see run)
    static constexpr int N = (NT && NB > 0 && NB < N0) ? NB : N0;
    static constexpr int B = N;
    T v[N][NV][VEC];
    int nvs[N];
    int64_t i0, step, end;
    const T *ptr[NV];

    __device__ __forceinline__ void issue(int64_t n, int ch, int c, const T *const (&p)[NV]) {
        const int64_t base = (int64_t)c * ch;
        end = (base + ch < n) ? base + ch : n;
        step = (int64_t)VEC * HIPK_THREADS;
        i0 = base + (int64_t)VEC * threadIdx.x;
#pragma unroll
        for (int a = 0; a < NV; ++a) ptr[a] = p[a];
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const int64_t i = i0 + k * step;
            nvs[k] = (i < end) ? ((end - i < VEC) ? (int)(end - i) : VEC) : 0;
            if (nvs[k] > 0) {
#pragma unroll
                for (int a = 0; a < NV; ++a) {
                    if (NT) hipk_ld_nt_vec<T>(ptr[a], i, nvs[k], v[k][a]);
                    else hipk_ld<T>(ptr[a], i, nvs[k], v[k][a]);
                }
            }
        }
    }
    template <typename F>
    __device__ __forceinline__ void run(F f) {
#pragma unroll
        for (int k = 0; k < N; ++k)
            if (nvs[k] > 0) f(i0 + k * step, nvs[k], v[k]);
        // larger chunks (N > 4 M rows): BATCHES of N steps through the same registers -- every load of a batch is issued before
        // the batch's first store (the kernels update their operands in place, so the compiler may not move a step's loads above
        // the previous step's stores).  Same-box A/B at N = 64 M (profiles/r02_vector_tail_ab.txt): CG update kernel 278 -> 272 us,
        // direction kernel 457 -> 464 us -- the round trips were already hidden by the 30 wavefronts per CU; kept for the 2 %.
        // Only with the streaming policy (NT: systems whose vectors live in HBM): the cache-resident sizes keep the step-by-step
        // tail, whose registers the hot N = 4 M instantiations are budgeted for (the batched form took them from 64 to 78 VGPRs).
        for (int64_t ib = i0 + N * step; NT && ib < end; ib += B * step) {
#pragma unroll
            for (int k = 0; k < B; ++k) {
                const int64_t i = ib + k * step;
                nvs[k] = (i < end) ? ((end - i < VEC) ? (int)(end - i) : VEC) : 0;
                if (nvs[k] > 0) {
#pragma unroll
                    for (int a = 0; a < NV; ++a) {
                        if (NT) hipk_ld_nt_vec<T>(ptr[a], i, nvs[k], v[k][a]);
                        else hipk_ld<T>(ptr[a], i, nvs[k], v[k][a]);
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < B; ++k)
                if (nvs[k] > 0) f(ib + k * step, nvs[k], v[k]);
        }
        for (int64_t i = i0 + N * step; !NT && i < end; i += step) {
            const int nv = (end - i < VEC) ? (int)(end - i) : VEC;
            T w[NV][VEC];
#pragma unroll
            for (int a = 0; a < NV; ++a) hipk_ld<T>(ptr[a], i, nv, w[a]);
            f(i, nv, w);
        }
    }
};
#endif

// out_dev[0] = fixed-order sum of part[0..g)   (single workgroup)
int hipk_launch_finish1(const double *part, int g, double *out_dev, hipStream_t stream);
// part[c] = chunk partial of <x,y>
int hipk_launch_dot_parts(int64_t n, const void *x, const void *y, int dtype, double *part,
                          hipStream_t stream);
// same with an explicit chunk size (row block of a larger, partitioned vector)
int hipk_launch_dot_parts_ch(int64_t n, int ch, const void *x, const void *y, int dtype, double *part,
                             hipStream_t stream);
