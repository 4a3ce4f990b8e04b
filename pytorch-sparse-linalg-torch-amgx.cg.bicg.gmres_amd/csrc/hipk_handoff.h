// hipk_handoff.h -- in-kernel hand-offs between the resident workgroups of the small-system "whole solve in one launch" kernels
// (hipk_gm_solve_lds_kernel, hipk_cg_solve_lds_kernel) and the sub-workgroup folds of the reduction spec they share.
#ifndef HIPK_HANDOFF_H
#define HIPK_HANDOFF_H
#include "hipk_common.h"

__device__ __forceinline__ void hipk_publish(double *p, double v) {
    __hip_atomic_store((unsigned long long *)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double hipk_peek(const double *p) {
    return __longlong_as_double((long long)__hip_atomic_load((const unsigned long long *)p, __ATOMIC_RELAXED,
                                                             __HIP_MEMORY_SCOPE_AGENT));
}

template <typename T>
__device__ __forceinline__ T hipk_peek_t(const T *p);
template <>
__device__ __forceinline__ double hipk_peek_t<double>(const double *p) { return hipk_peek(p); }
template <>
__device__ __forceinline__ float hipk_peek_t<float>(const float *p) {
    return __uint_as_float(__hip_atomic_load((const unsigned *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void hipk_publish_t(double *p, double v) { hipk_publish(p, v); }
__device__ __forceinline__ void hipk_publish_t(float *p, float v) {
    __hip_atomic_store((unsigned *)p, __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// counter barrier of the `nwg` resident workgroups; false when the spin bound was hit (another workgroup never arrived)
__device__ __forceinline__ bool hipk_gbar(int32_t *ctr, int nwg, int &epoch, int *fail_lds) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wavefront: its (sc1) stores have left
    __syncthreads();
    ++epoch;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int target = epoch * nwg;
        unsigned spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 20)) {  // ~a second: a workgroup of this launch is not running
                *fail_lds = 1;
                break;
            }
        }
    }
    __syncthreads();
    return *fail_lds == 0;
}

template <typename T, class F>
__device__ __forceinline__ double hipk_fold_8x8(int i8, int g, F val) {
    // lanes i8 = 0..7 of an 8-lane group: inner fold of val(i8, 0..7), then the spec's fold of <= 8 chunk partials across
    // the group (row_shl 4, 2, 1); valid where i8 == 0.  All 8 lanes must be active.
    double a = 0.0;
    if (i8 < g) {
        double p[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) p[s] = val(i8, s);
        a = 0.0 + (((p[0] + p[4]) + (p[2] + p[6])) + ((p[1] + p[5]) + (p[3] + p[7])));
    } else {
        a = 0.0 + 0.0;
    }
    a = a + hipk_row_shl<4>(a);
    a = a + hipk_row_shl<2>(a);
    a = a + hipk_row_shl<1>(a);
    return a;
}
// The same for up to 64 chunks, one LANE per chunk (a whole wavefront must call): inner fold of val(lane, 0..7), then the spec's
// fold of the chunk results -- thread t takes partial t (g <= 256: one each), tree v[t] += v[t+s], of which the strides 32 .. 1
// reach the <= 64 non-zero slots.  For g <= 8 the additions (and bits) are those of hipk_fold_8x8.  Valid in lane 0.
template <class F>
__device__ __forceinline__ double hipk_fold_64x8(int lane, int g, F val) {
    double a = 0.0;
    if (lane < g) {
        double p[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) p[s] = val(lane, s);
        a = 0.0 + (((p[0] + p[4]) + (p[2] + p[6])) + ((p[1] + p[5]) + (p[3] + p[7])));
    } else {
        a = 0.0 + 0.0;
    }
    return hipk_wave_sum(a);
}
static constexpr int kHoMaxChunks = 64;    // chunks (lanes of the fold above) the one-launch kernels handle: n <= 131072
static constexpr int kHoMaxWg = 8 * kHoMaxChunks;
// sum over the 32 lanes of a half wavefront with the strides 16 .. 1 of the spec's tree; valid in lanes 0 and 32
__device__ __forceinline__ double hipk_half_sum(double d) {
    d = d + hipk_lane_up16(d);
    d = d + hipk_row_shl<8>(d);
    d = d + hipk_row_shl<4>(d);
    d = d + hipk_row_shl<2>(d);
    d = d + hipk_row_shl<1>(d);
    return d;
}

static constexpr int kGmSub = 8;        // sub-workgroups per reduction chunk

// ---- hand-offs between the workgroups of hipk_gm_solve_lds_kernel
// LOCAL = true: every workgroup runs on the SAME XCD (verified at kernel start from HW_REG_XCC_ID, else the launch gives up):
// that XCD's L2 is their coherence point, so payload and flags are PLAIN stores (the lines stay in L2) read with sc1 loads
// (which only bypass the reader's L1): a hand-off costs L2 round trips.  LOCAL = false: agent-scope (sc1, write-through)
// stores, valid on any placement, every trip through the fabric.  Each workgroup owns one flag word per hand-off kind and
// stores the hand-off's sequence number into it after ALL its waves have drained their stores; a consumer polls the
// flags with one wave-wide load per 64 workgroups.  Up to 64 workgroups (n <= 16384) run on ONE XCD; larger systems spread over the
// chip (one workgroup per block, agent-scope hand-offs only).
template <bool LOCAL>
__device__ __forceinline__ void hipk_ho_store(double *p, double v) {
    if (LOCAL)
        __hip_atomic_store((unsigned long long *)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    else
        hipk_publish(p, v);
}
template <bool LOCAL>
__device__ __forceinline__ void hipk_ho_store(float *p, float v) {
    if (LOCAL)
        __hip_atomic_store((unsigned *)p, __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    else
        hipk_publish_t(p, v);
}
template <bool LOCAL>
__device__ __forceinline__ void hipk_ho_flag(unsigned long long *p, unsigned long long v) {
    if (LOCAL)
        __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    else
        __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// signal hand-off number `seq` (flag value 2 seq + bit) for this workgroup, then wait until every workgroup has signalled it.
// Returns the flag word of workgroup 0 (its low bit carries the stop decision), or ~0 when the spin bound was hit.
struct hipk_no_side_work {
    __device__ __forceinline__ void operator()() const {}
};
// `side`: work of thread 192 (wavefront 3 only waits here) that the consumers of this hand-off need afterwards
template <bool LOCAL, class F = hipk_no_side_work>
__device__ __forceinline__ unsigned long long hipk_ho_sync(unsigned long long *flags, int wg, int nwg, unsigned long long seq,
                                                           unsigned bit, unsigned long long *res_lds, F side = F()) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wavefront: its stores have arrived
    __syncthreads();
    if (threadIdx.x == 192) side();
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        if (lane == 0) hipk_ho_flag<LOCAL>(flags + wg, 2 * seq + bit);
        unsigned long long f0 = ~0ull;
        unsigned spins = 0;
        for (;;) {   // <= 512 flags: one wave-wide load per 64
            bool all = true;
            unsigned long long first = 0;
            for (int j0 = 0; j0 < nwg; j0 += 64) {
                const unsigned long long f = (j0 + lane < nwg) ? __hip_atomic_load(flags + j0 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ~0ull;
                all = all && __all(f >= 2 * seq);
                if (j0 == 0) first = __shfl(f, 0);
            }
            if (all) {
                f0 = first;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 20)) break;   // ~a second: a workgroup of this launch is not running
        }
        if (lane == 0) *res_lds = f0;
    }
    __syncthreads();
    return *res_lds;
}

// sc1 load of base[byte_off / sizeof(T)]: an SGPR base + 32-bit VGPR offset, so that gathers through several bases (column 0,
// a.q, x) do not each keep sixteen 64-bit addresses alive
template <typename T>
__device__ __forceinline__ T hipk_peek_off(const T *base, unsigned byte_off) {
    return hipk_peek_t<T>((const T *)((const char *)base + (size_t)byte_off));
}

#endif  // HIPK_HANDOFF_H
