// hipk_gmres.hip -- restarted GMRES with the Krylov basis resident in HBM.
//
// Restates `gmres` (TSL:641-803) with `_gmres_batched` (TSL:431-493) and
// `_gmres_incremental` (TSL:557-638) for M = identity.  What changes against the
// reference's data layout and launch pattern:
//   * the basis is COLUMN-CONTIGUOUS (column j = one aligned n-vector); the reference keeps
//     the basis index fastest (TSL:448-452) and CLONES the whole n x (m+1) basis every
//     Arnoldi step (TSL:363-368) -- here A v_k is written straight into column k+1 and
//     orthogonalised / normalised in place;
//   * one Arnoldi step = SpMV (+||w||^2) | multi-dot h = V^H w over the k+1 LIVE columns
//     (one pass over w, k+1 accumulators per thread) | tiny reduce | fused q = w - V h
//     (+||q||^2) | CGS2 decision on the device (TSL:322-326) | optional second pass |
//     guarded normalise + Hessenberg column (+ Givens update for 'incremental');
//   * no host synchronisation inside a restart cycle: breakdown (TSL:387) and the
//     incremental early exit (TSL:591, 622) set a device stop word that turns the
//     remaining steps into no-ops; the host reads H once per cycle and solves the
//     <= 31 x 30 least-squares problem (normal equations + Cholesky, TSL:407-421, or the
//     triangular solve TSL:630) in the same arithmetic order as the oracle.
// Traffic of step k (fp64): B_spmv + 8n(k+2) [multi-dot] + 8n(k+3) [update] + 16n [normalise].
#include <math.h>
#include <stdlib.h>

#include "hipk_blas1.h"
#include "hipk_solve.h"
#include "hipk_spmv.h"
#include "hipk_handoff.h"

#define HIPK_GM_MAXM 31        // restart bound of the small-system (LDS / one-launch) kernels and of the arrays INSIDE hipk_gm_scal
#define HIPK_GM_LDH 32
#define HIPK_GM_MAXM_BIG 255   // restart bound of the launch sequences: beyond 31 the Hessenberg arrays live in the workspace

// Where the m-dependent small arrays of a cycle live: inside hipk_gm_scal (restart <= 31, leading dimension 32) or in a block of
// the workspace sized for the solve's restart (hipk_gm_big_doubles).  Filled by hipk_gm_cycle_init_kernel; the kernels of the
// launch sequences address H, R, ... through it, the small-system kernels (restart <= 31 only) use the struct's arrays directly.
struct hipk_gm_view {
    double *H;         // (m+2) x ldh, row-major
    double *R;         // ldh x ldh
    double *gv;        // 2 ldh
    double *beta_vec;  // ldh + 1
    double *hvec, *rvec;  // ldh each
    int ldh, m;
};
static inline int hipk_gm_big_ld(int m) { return ((m + 1 + 7) / 8) * 8; }
static inline size_t hipk_gm_big_doubles(int m) {   // 0 for restart <= 31
    if (m <= HIPK_GM_MAXM) return 0;
    const size_t ld = (size_t)hipk_gm_big_ld(m);
    return (((size_t)(m + 2) * ld + ld * ld + 2 * ld + (ld + 1) + 2 * ld) + 31) / 32 * 32;
}
#define HIPK_EPS64 2.220446049250313e-16
#define HIPK_EPS32 1.1920928955078125e-07
#define HIPK_INV_SQRT2 0.7071067811865476

struct hipk_gm_scal {
    double res_norm;  // ||r|| after `_safe_normalize` (0 when <= eps)
    double bs;        // <b,b>
    double res2, xx;  // final true residual^2 and <x,x>
    double qnorm;     // step-local ||q|| after CGS pass 1 (thresholded)
    double err;       // incremental: |beta_vec[k+1]|
    double ptol;
    int64_t stop_step;  // Arnoldi steps >= stop_step of the current cycle are no-ops
    int64_t steps_done;
    int32_t pass2;      // second CGS pass wanted for the current step
    int32_t breakdown;
    int32_t incremental;
    int32_t pad;
    double H[(HIPK_GM_MAXM + 2) * HIPK_GM_LDH];  // (m+1) x m, row-major, ld 32
    double R[HIPK_GM_LDH * HIPK_GM_LDH];
    double gv[HIPK_GM_LDH * 2];
    double beta_vec[HIPK_GM_LDH + 1];
    double hvec[HIPK_GM_LDH];
    double rvec[HIPK_GM_LDH];
    int32_t bar;        // counter barrier of hipk_gm_cycle_small_kernel (zeroed by hipk_gm_cycle_init_kernel)
    int32_t redo;       // speculation miss: a second CGS pass was wanted at a step whose pass-2 launches were not enqueued
    int64_t redo_step;
    // hipk_gm_solve_lds_kernel: per-workgroup hand-off flags and the XCDs its workgroups found themselves on
    unsigned long long flag_md[64];
    unsigned long long flag_q[64];
    unsigned xcc_mask;
    unsigned pad2;
    // what a launch of hipk_gm_solve_lds_kernel reports
    long long rep_cycles;   // restart cycles finished by the launch
    long long rep_matvecs;  // operator applications of the launch
    int rep_status;         // 0: cycle budget of the launch used up; 1: converged or out of cycles; 2: the Cholesky factorisation
                            // of the current cycle's normal equations failed -- H, steps_done and the basis are in memory, the
                            // host finishes this cycle (general solve, TSL:424-428)
    int rep_breakdown;      // a breakdown (TSL:387) happened in one of the launch's cycles
    hipk_gm_view v;         // see above (set by hipk_gm_cycle_init_kernel)
    double vnorm;           // hipk_gm_hcol_kernel -> hipk_gm_scale_kernel: ||q|| of the step ...
    int32_t vuse, pad3;     // ... and whether it passed the threshold (else v_{k+1} = 0)
};
static constexpr size_t kGmHeader = 32768;
static_assert(sizeof(hipk_gm_scal) <= kGmHeader, "header too small");
static constexpr int kGmSlots = 8;  // ww, qq, res, bb, xx, spare x3
static constexpr int kGmSplitChunks = 1536;  // normalise step as two launches from this many reduction chunks (see hipk_gm_hcol_kernel):
                                             // GMRES(30) ms per cycle, split / one kernel: 44 chunks 1.34-1.36 / 1.31, 123: 1.56 / 1.51, 254: 1.83-1.86 / 1.80,
                                             // 489: 2.41 / 2.38, 958: 3.74 / 3.68-3.70, 1954: 6.54-6.55 / 6.61 (profiles/r03_gmres_history.md)

template <int KC>
struct hipk_gm_yN {
    double y[KC];
};
typedef hipk_gm_yN<HIPK_GM_LDH> hipk_gm_y;

// ---- reduce 8 per-thread values at once with the spec tree (sbuf: 8*256 doubles)
__device__ __forceinline__ void hipk_block_sum8(double (&v)[8], int nb, double *sbuf) {
    const int t = threadIdx.x;
#pragma unroll
    for (int b = 0; b < 8; ++b) sbuf[b * HIPK_THREADS + t] = v[b];
    __syncthreads();
    if (t < 128) {
#pragma unroll
        for (int b = 0; b < 8; ++b)
            sbuf[b * HIPK_THREADS + t] = sbuf[b * HIPK_THREADS + t] + sbuf[b * HIPK_THREADS + t + 128];
    }
    __syncthreads();
    if (t < 64) {
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            double a = sbuf[b * HIPK_THREADS + t] + sbuf[b * HIPK_THREADS + t + 64];
a = hipk_wave_sum(a);
            v[b] = a;  // valid in lane 0
        }
    }
    __syncthreads();
    (void)nb;
}

// Small systems (at most 8 reduction chunks, n <= 16384): an Arnoldi step is launch-bound (10 launches of ~4 us for a few
// KB of data), so three of them are folded into their consumers.  hipk_fold8 is the spec's fold of g <= 8 partials
// (hipk_reduce_parts: acc = 0.0 + part[t], then the tree v[t] += v[t+s], s = 128..1, of which only s = 4, 2, 1 touch
// non-zero slots) evaluated by ONE thread: same additions, same order, same bits.
__device__ __forceinline__ double hipk_fold8(const double *__restrict__ part, int g) {
    double a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = 0.0 + ((i < g) ? part[i] : 0.0);
    return ((a[0] + a[4]) + (a[2] + a[6])) + ((a[1] + a[5]) + (a[3] + a[7]));
}

// second CGS pass iff ||r|| < ||q|| / sqrt(2)  (TSL:313-326); shared by hipk_gm_decide_kernel and the SMALL multi-dot
__device__ __forceinline__ int hipk_gm_want_pass2(const hipk_gm_scal *scal, int k, double qq, double eps, double *qnorm_out,
                                                  const double *rvec = nullptr) {
    if (rvec == nullptr) rvec = scal->rvec;   // small-system kernels: the struct's own array
    double qnorm = sqrt(qq < 0.0 ? 0.0 : qq);
    if (!(qnorm > eps)) qnorm = 0.0;
    double rr = 0.0;
    for (int j = 0; j <= k; ++j) rr = fma(rvec[j], rvec[j], rr);
    double rnorm = sqrt(rr < 0.0 ? 0.0 : rr);
    if (!(rnorm > eps)) rnorm = 0.0;
    *qnorm_out = qnorm;
    return (rnorm < qnorm * HIPK_INV_SQRT2) ? 1 : 0;
}

// part[j*MAXP + c] = chunk partial of <V_j, w>, j = 0..k  (`_project_on_columns`, TSL:276-281)
// grid = (chunks, ceil((k+1)/8)): a workgroup takes EIGHT columns of one chunk -- eight accumulators and eight
// column streams per thread keep it at 8 workgroups per CU (31 accumulators in one workgroup: 2-4 per CU, several
// rounds of workgroups); w's chunk is re-read by each column group from L2.  All eight loads of a step are issued
// before their FMAs.  Per column the accumulation order is the spec's.
// SMALL (g <= 8): the pass-2 launch takes the CGS2 decision itself (no hipk_gm_decide_kernel launch): every workgroup
// derives it from the same partials, workgroup (0,0) publishes it for the kernels that follow.
template <typename T, bool SMALL>
__global__ __launch_bounds__(HIPK_THREADS) void hipk_gm_multidot_kernel(
    int64_t n, int ch, hipk_gm_scal *__restrict__ scal, int k, int pass, const T *__restrict__ V, int64_t ldv,
    const T *__restrict__ w, double *__restrict__ part, const double *__restrict__ part_qq, int g, double eps) {
    if (k >= scal->stop_step) return;
    if (pass == 1) {
        if (SMALL) {
            double qnorm;
            const int want = hipk_gm_want_pass2(scal, k, hipk_fold8(part_qq, g), eps, &qnorm);
            if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
                scal->qnorm = qnorm;
                scal->pass2 = want;
            }
            if (!want) return;
        } else if (!scal->pass2) {
            return;
        }
    }
    __shared__ double sbuf[8 * HIPK_THREADS];
    const int c = blockIdx.x;
    const int j0 = 8 * blockIdx.y;  // <= k by construction of the grid
    double acc[8];
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[b] = 0.0;
    hipk_chunk_loop<T, 1>(n, ch, c, [&](int64_t i, int nv) {
        constexpr int VEC = hipk_vec<T>::VEC;
        T wv[VEC];
        hipk_ld<T>(w, i, nv, wv);
        T vv[8][VEC];
#pragma unroll
        for (int b = 0; b < 8; ++b)
            if (j0 + b <= k) hipk_ld<T>(V + (int64_t)(j0 + b) * ldv, i, nv, vv[b]);
#pragma unroll
        for (int b = 0; b < 8; ++b)
            if (j0 + b <= k) {
#pragma unroll
                for (int e = 0; e < VEC; ++e)
                    if (e < nv) acc[b] = fma((double)vv[b][e], (double)wv[e], acc[b]);
            }
    });
    hipk_block_sum8(acc, 8, sbuf);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int b = 0; b < 8; ++b)
            if (j0 + b <= k) part[(size_t)(j0 + b) * HIPK_MAX_PARTS + c] = acc[b];
    }
}

// Small systems, chunk = HIPK_BASE_CHUNK: the update kernel above is one 256-thread workgroup per chunk walking its
// 4 (fp64) / 2 (fp32) chunk-loop steps one after the other, each a dependent round of column loads -- 8 + 0.43 k us at
// n = 10^4 with five CUs busy.  Here the workgroup has one 256-thread group PER STEP (1024 / 512 threads): all column
// loads of the chunk are in flight at once.  Arithmetic per element is unchanged; the per-thread <w,w> chain of the
// spec (step 0's elements first, then step 1's, ...) is handed from group to group through LDS.
template <typename T>
__global__ __launch_bounds__(HIPK_BASE_CHUNK / hipk_vec<T>::VEC) void hipk_gm_update_wide_kernel(
    int64_t n, hipk_gm_scal *__restrict__ scal, int k, int pass, const T *__restrict__ V, int64_t ldv,
    T *__restrict__ w, double *__restrict__ part_qq, const double *__restrict__ part_md, int g) {
    constexpr int VEC = hipk_vec<T>::VEC;
    constexpr int NIT = HIPK_BASE_CHUNK / (VEC * HIPK_THREADS);
    if (k >= scal->stop_step) return;
    if (pass == 1 && !scal->pass2) return;
    __shared__ double chain[HIPK_THREADS];
    __shared__ double hs[HIPK_GM_LDH];
    const int t = threadIdx.x & (HIPK_THREADS - 1), q = threadIdx.x / HIPK_THREADS;
    if (threadIdx.x < HIPK_GM_LDH) {
        double hj = 0.0;
        if (threadIdx.x <= k) hj = hipk_fold8(part_md + (size_t)threadIdx.x * HIPK_MAX_PARTS, g);
        hs[threadIdx.x] = hj;
    }
    __syncthreads();
    const int c = blockIdx.x;
    const int64_t base = (int64_t)c * HIPK_BASE_CHUNK;
    const int64_t end = (base + HIPK_BASE_CHUNK < n) ? base + HIPK_BASE_CHUNK : n;
    const int64_t i = base + (int64_t)VEC * t + (int64_t)q * VEC * HIPK_THREADS;
    const int nv = (i < end) ? ((end - i < VEC) ? (int)(end - i) : VEC) : 0;
    T wv[VEC];
    if (nv > 0) {
        hipk_ld<T>((const T *)w, i, nv, wv);
        double sacc[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) sacc[e] = 0.0;
#pragma unroll
        for (int j0 = 0; j0 < HIPK_GM_LDH; j0 += 8) {
            if (j0 <= k) {
                T vv[8][VEC];
#pragma unroll
                for (int b = 0; b < 8; ++b)
                    if (j0 + b <= k) hipk_ld<T>(V + (int64_t)(j0 + b) * ldv, i, nv, vv[b]);
#pragma unroll
                for (int b = 0; b < 8; ++b)
                    if (j0 + b <= k) {
                        const double hj = hs[j0 + b];
#pragma unroll
                        for (int e = 0; e < VEC; ++e) sacc[e] = fma((double)vv[b][e], hj, sacc[e]);
                    }
            }
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) wv[e] = (T)((double)wv[e] - sacc[e]);
        hipk_st<T>(w, i, nv, wv);
    }
#pragma unroll
    for (int qq = 0; qq < NIT; ++qq) {  // the spec's per-thread chain, one chunk-loop step per group
        if (q == qq) {
            double a = (qq == 0) ? 0.0 : chain[t];
#pragma unroll
            for (int e = 0; e < VEC; ++e)
                if (e < nv) a = fma((double)wv[e], (double)wv[e], a);
            chain[t] = a;
        }
        __syncthreads();
    }
    // the spec's tree over the 256 chains (hipk_block_sum), every thread of the wide workgroup at the barriers
    if (threadIdx.x < 128) chain[threadIdx.x] = chain[threadIdx.x] + chain[threadIdx.x + 128];
    __syncthreads();
    if (threadIdx.x < 64) {
        double a = chain[threadIdx.x] + chain[threadIdx.x + 64];
a = hipk_wave_sum(a);
        if (threadIdx.x == 0) {
            part_qq[c] = a;
            if (c == 0) {
                for (int j = 0; j <= k; ++j) scal->rvec[j] = ((pass == 0) ? 0.0 : scal->rvec[j]) + hs[j];
            }
        }
    }
}

// The multidot of small systems in the same wide form (one 256-thread group per chunk-loop step, 8 chains per thread
// handed on through LDS).
template <typename T>
__global__ __launch_bounds__(HIPK_BASE_CHUNK / hipk_vec<T>::VEC) void hipk_gm_multidot_wide_kernel(
    int64_t n, hipk_gm_scal *__restrict__ scal, int k, int pass, const T *__restrict__ V, int64_t ldv,
    const T *__restrict__ w, double *__restrict__ part, const double *__restrict__ part_qq, int g, double eps) {
    constexpr int VEC = hipk_vec<T>::VEC;
    constexpr int NIT = HIPK_BASE_CHUNK / (VEC * HIPK_THREADS);
    const int c = blockIdx.x;
    const int j0 = 8 * blockIdx.y;  // <= k by construction of the grid
    const int t = threadIdx.x & (HIPK_THREADS - 1), q = threadIdx.x / HIPK_THREADS;
    const int64_t base = (int64_t)c * HIPK_BASE_CHUNK;
    const int64_t end = (base + HIPK_BASE_CHUNK < n) ? base + HIPK_BASE_CHUNK : n;
    const int64_t i = base + (int64_t)VEC * t + (int64_t)q * VEC * HIPK_THREADS;
    const int nv = (i < end) ? ((end - i < VEC) ? (int)(end - i) : VEC) : 0;
    if (k >= scal->stop_step) return;
    if (pass == 1) {
        double qnorm;
        const int want = hipk_gm_want_pass2(scal, k, hipk_fold8(part_qq, g), eps, &qnorm);
        if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
            scal->qnorm = qnorm;
            scal->pass2 = want;
        }
        if (!want) return;
    }
    __shared__ double chain[8 * HIPK_THREADS];
    T wv[VEC];
    T vv[8][VEC];
    if (nv > 0) {
        hipk_ld<T>(w, i, nv, wv);
#pragma unroll
        for (int b = 0; b < 8; ++b)
            if (j0 + b <= k) hipk_ld<T>(V + (int64_t)(j0 + b) * ldv, i, nv, vv[b]);
    }
#pragma unroll
    for (int qq = 0; qq < NIT; ++qq) {
        if (q == qq) {
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                double a = (qq == 0) ? 0.0 : chain[b * HIPK_THREADS + t];
                if (j0 + b <= k) {
#pragma unroll
                    for (int e = 0; e < VEC; ++e)
                        if (e < nv) a = fma((double)vv[b][e], (double)wv[e], a);
                }
                chain[b * HIPK_THREADS + t] = a;
            }
        }
        __syncthreads();
    }
    // hipk_block_sum8's tree, every thread of the wide workgroup at the barriers
    if (threadIdx.x < 128) {
#pragma unroll
        for (int b = 0; b < 8; ++b)
            chain[b * HIPK_THREADS + threadIdx.x] = chain[b * HIPK_THREADS + threadIdx.x] + chain[b * HIPK_THREADS + threadIdx.x + 128];
    }
    __syncthreads();
    if (threadIdx.x < 64) {
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            double a = chain[b * HIPK_THREADS + threadIdx.x] + chain[b * HIPK_THREADS + threadIdx.x + 64];
a = hipk_wave_sum(a);
            if (threadIdx.x == 0 && j0 + b <= k) part[(size_t)(j0 + b) * HIPK_MAX_PARTS + c] = a;
        }
    }
}

// hvec[j] = fixed-order sum of part[j][0..g)   (one workgroup per j)
__global__ __launch_bounds__(HIPK_THREADS) void hipk_gm_hreduce_kernel(hipk_gm_scal *__restrict__ scal, int k, int pass,
                                                                       int g, const double *__restrict__ part) {
    if (k >= scal->stop_step) return;
    if (pass == 1 && !scal->pass2) return;
    __shared__ double sbuf[HIPK_THREADS];
    const int j = blockIdx.x;
    const double h = hipk_reduce_parts(part + (size_t)j * HIPK_MAX_PARTS, g, sbuf);
    if (threadIdx.x == 0) scal->v.hvec[j] = h;
}

// q = w - V h in place, partials of <q,q>; rvec += h   (TSL:302-305)
// SMALL (g <= 8): h_j is folded here from the multi-dot partials (no hipk_gm_hreduce_kernel launch)
template <typename T, bool SMALL>
__global__ __launch_bounds__(HIPK_THREADS) void hipk_gm_update_kernel(
    int64_t n, int ch, hipk_gm_scal *__restrict__ scal, int k, int pass, const T *__restrict__ V, int64_t ldv,
    T *__restrict__ w, double *__restrict__ part_qq, const double *__restrict__ part_md, int g) {
    if (k >= scal->stop_step) return;
    if (pass == 1 && !scal->pass2) return;
    __shared__ double sbuf[HIPK_THREADS];
    __shared__ double hs[HIPK_GM_LDH];
    if (threadIdx.x < HIPK_GM_LDH) {
        double hj = 0.0;
        if (threadIdx.x <= k)
            hj = SMALL ? hipk_fold8(part_md + (size_t)threadIdx.x * HIPK_MAX_PARTS, g) : scal->v.hvec[threadIdx.x];
        hs[threadIdx.x] = hj;
    }
    __syncthreads();
    const int c = blockIdx.x;
    double acc = 0.0;
    hipk_chunk_loop<T, 1>(n, ch, c, [&](int64_t i, int nv) {
        constexpr int VEC = hipk_vec<T>::VEC;
        T wv[VEC];
        hipk_ld<T>((const T *)w, i, nv, wv);
        double s[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) s[e] = 0.0;
#pragma unroll
        for (int j0 = 0; j0 < HIPK_GM_LDH; j0 += 8) {  // batches of eight column loads, then their FMAs in column order
            if (j0 <= k) {
                T vv[8][VEC];
#pragma unroll
                for (int b = 0; b < 8; ++b)
                    if (j0 + b <= k) hipk_ld<T>(V + (int64_t)(j0 + b) * ldv, i, nv, vv[b]);
#pragma unroll
                for (int b = 0; b < 8; ++b)
                    if (j0 + b <= k) {
                        const double hj = hs[j0 + b];
#pragma unroll
                        for (int e = 0; e < VEC; ++e) s[e] = fma((double)vv[b][e], hj, s[e]);
                    }
            }
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            wv[e] = (T)((double)wv[e] - s[e]);
            if (e < nv) acc = fma((double)wv[e], (double)wv[e], acc);
        }
        hipk_st<T>(w, i, nv, wv);
    });
    acc = hipk_block_sum(acc, sbuf);
    if (threadIdx.x == 0) {
        part_qq[c] = acc;
        if (c == 0) {
            double *rvec = SMALL ? scal->rvec : scal->v.rvec;
            for (int j = 0; j <= k; ++j) rvec[j] = ((pass == 0) ? 0.0 : rvec[j]) + hs[j];
        }
    }
}

// second CGS pass iff ||r|| < ||q|| / sqrt(2)  (TSL:313-326), norms guarded as `_safe_normalize`
__global__ __launch_bounds__(HIPK_THREADS) void hipk_gm_decide_kernel(hipk_gm_scal *__restrict__ scal, int k, int g,
                                                                      const double *__restrict__ part_qq, double eps) {
    if (k >= scal->stop_step) return;
    __shared__ double sbuf[HIPK_THREADS];
    const double qq = hipk_reduce_parts(part_qq, g, sbuf);
    if (threadIdx.x == 0) {
        double qnorm;
        scal->pass2 = hipk_gm_want_pass2(scal, k, qq, eps, &qnorm, scal->v.rvec);
        scal->qnorm = qnorm;
    }
}


// =====================================================================================================================
// Large systems: the Krylov basis (31 x 32 MB at N = 4 M) is far larger than the 256 MiB Infinity Cache and every Arnoldi
// step reads its live columns twice (h = V^H w, then q = w - V h, with a global reduction in between).  Measured on
// MI355X (profiles/r02_gmres_history.md):
//  * STREAMING POLICY.  Basis columns >= `nres` are loaded NON-TEMPORAL: they pass through without displacing what is
//    re-read soon (w, the first `nres` columns, the SpMV's operands).  8.15 -> 6.80 ms per GMRES(30) cycle at N = 4 M.
//  * tried and rejected: chunk-major sweeps with alternating direction (the turn-around hits of the Infinity Cache bought
//    nothing measurable: MALL-served reads are only ~1.4x faster than HBM reads) and last-arriver folds of the partial sums
//    inside the kernels (ticket atomics + write-through stores per workgroup: +1.3 ms per cycle).
// The order of operations per element (columns ascending) is the spec's: same bits as the kernels above.
// NC-column tree of the spec (hipk_block_sum8 for NC <= 8 live columns: no LDS traffic for dead ones)
template <int NC>
__device__ __forceinline__ void hipk_block_sumN(double (&v)[NC], double *sbuf) {
    const int t = threadIdx.x;
#pragma unroll
    for (int b = 0; b < NC; ++b) sbuf[b * HIPK_THREADS + t] = v[b];
    __syncthreads();
    if (t < 128) {
#pragma unroll
        for (int b = 0; b < NC; ++b)
            sbuf[b * HIPK_THREADS + t] = sbuf[b * HIPK_THREADS + t] + sbuf[b * HIPK_THREADS + t + 128];
    }
    __syncthreads();
    if (t < 64) {
#pragma unroll
        for (int b = 0; b < NC; ++b) {
            double a = sbuf[b * HIPK_THREADS + t] + sbuf[b * HIPK_THREADS + t + 64];
            a = hipk_wave_sum(a);
            v[b] = a;  // valid in lane 0
        }
    }
    __syncthreads();
}

// part[j*MAXP + c] = chunk partial of <V_j, w>.  grid = g * (k / GW + 1), GW = max(8, NC): workgroup b takes chunk b % g of
// column group b / g.  NC = compile-time bound on the live columns of a group: 1, 2, 4, 8 for the latency-bound first steps (with
// NC <= 2 the chunk's four steps are all in flight at once and the tree only carries live columns), 16 and 32 beyond -- ONE group
// up to k = 31, so w is read once per step (round 2 took groups of 8 throughout: w re-read per group, PMC 1.15 x the algorithmic
// bytes over a GMRES(30) cycle).  The column loads still go out in batches of eight (8 x 16 B per lane in flight) and the block
// tree runs per batch of eight accumulators (16 KB of LDS whatever NC).  Per column the accumulation order is the spec's.
template <typename T, int NC>
__global__ __launch_bounds__(HIPK_THREADS) HIPK_SGPR80 void hipk_gm_multidot_stream_kernel(
    int64_t n, int ch, hipk_gm_scal *__restrict__ scal, int k, int pass, const T *__restrict__ V, int64_t ldv,
    const T *__restrict__ w, double *__restrict__ part, int g, int nres) {
    if (k >= scal->stop_step) return;
    if (pass == 1 && !scal->pass2) return;
    constexpr int GW = NC < 8 ? 8 : NC;   // columns per group
    constexpr int NB = NC < 8 ? NC : 8;   // columns per load batch / tree batch
    const int grp = blockIdx.x / g, c = blockIdx.x % g;
    __shared__ double sbuf[NB * HIPK_THREADS];
    const int j0 = GW * grp;
    double acc[NC];
#pragma unroll
    for (int b = 0; b < NC; ++b) acc[b] = 0.0;
    hipk_chunk_loop<T, (NC <= 2 ? 4 : 1)>(n, ch, c, [&](int64_t i, int nv) {
        constexpr int VEC = hipk_vec<T>::VEC;
        T wv[VEC];
        hipk_ld<T>(w, i, nv, wv);
#pragma unroll
        for (int b0 = 0; b0 < NC; b0 += NB) {
            if (j0 + b0 <= k) {   // uniform
                T vv[NB][VEC];
#pragma unroll
                for (int b = 0; b < NB; ++b)
                    if (j0 + b0 + b <= k) {
                        if (j0 + b0 + b < nres) hipk_ld<T>(V + (int64_t)(j0 + b0 + b) * ldv, i, nv, vv[b]);
                        else hipk_ld_nt_vec<T>(V + (int64_t)(j0 + b0 + b) * ldv, i, nv, vv[b]);
                    }
#pragma unroll
                for (int b = 0; b < NB; ++b)
                    if (j0 + b0 + b <= k) {
#pragma unroll
                        for (int e = 0; e < VEC; ++e)
                            if (e < nv) acc[b0 + b] = fma((double)vv[b][e], (double)wv[e], acc[b0 + b]);
                    }
            }
        }
    });
#pragma unroll
    for (int b0 = 0; b0 < NC; b0 += NB) {
        if (j0 + b0 <= k) {   // uniform
            double a8[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) a8[b] = acc[b0 + b];
            hipk_block_sumN<NB>(a8, sbuf);
            if (threadIdx.x == 0) {
#pragma unroll
                for (int b = 0; b < NB; ++b)
                    if (j0 + b0 + b <= k) part[(size_t)(j0 + b0 + b) * HIPK_MAX_PARTS + c] = a8[b];
            }
        }
    }
}

// q = w - V h in place, partials of <q,q>; rvec += h   (TSL:302-305).  grid = g.
// KC = compile-time bound on the live columns (8: the first eight Arnoldi steps, one round of column loads and few enough
// registers for 8 workgroups per CU -- those steps are latency-bound; 32: the general form; 256: restart > 31, batches in a run-time loop).
template <typename T, int KC>
__global__ __launch_bounds__(HIPK_THREADS, (KC <= 8 ? 8 : 1)) void hipk_gm_update_stream_kernel(
    int64_t n, int ch, hipk_gm_scal *__restrict__ scal, int k, int pass, const T *__restrict__ V, int64_t ldv,
    T *__restrict__ w, double *__restrict__ part_qq, int nres) {
    if (k >= scal->stop_step) return;
    if (pass == 1 && !scal->pass2) return;
    const int c = blockIdx.x;
    __shared__ double sbuf[HIPK_THREADS];
    constexpr int HS = KC < HIPK_GM_LDH ? HIPK_GM_LDH : KC;
    __shared__ double hs[HS];
    if (threadIdx.x < HS) hs[threadIdx.x] = (threadIdx.x <= k) ? scal->v.hvec[threadIdx.x] : 0.0;
    __syncthreads();
    double acc = 0.0;
    hipk_chunk_loop<T, 1>(n, ch, c, [&](int64_t i, int nv) {
        constexpr int VEC = hipk_vec<T>::VEC;
        T wv[VEC];
        hipk_ld<T>((const T *)w, i, nv, wv);
        double s[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) s[e] = 0.0;
        constexpr int kUnroll = KC <= HIPK_GM_LDH ? KC / 8 : 1;
#pragma unroll kUnroll   // restart > 31: a run-time loop over the batches (60 VGPRs instead of 190-256)
        for (int j0 = 0; j0 < KC; j0 += 8) {  // batches of eight column loads, then their FMAs in column order
            if (j0 <= k) {
                T vv[8][VEC];
#pragma unroll
                for (int b = 0; b < 8; ++b)
                    if (j0 + b <= k) {
                        if (j0 + b < nres) hipk_ld<T>(V + (int64_t)(j0 + b) * ldv, i, nv, vv[b]);
                        else hipk_ld_nt_vec<T>(V + (int64_t)(j0 + b) * ldv, i, nv, vv[b]);
                    }
#pragma unroll
                for (int b = 0; b < 8; ++b)
                    if (j0 + b <= k) {
                        const double hj = hs[j0 + b];
#pragma unroll
                        for (int e = 0; e < VEC; ++e) s[e] = fma((double)vv[b][e], hj, s[e]);
                    }
            }
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            wv[e] = (T)((double)wv[e] - s[e]);
            if (e < nv) acc = fma((double)wv[e], (double)wv[e], acc);
        }
        hipk_st<T>(w, i, nv, wv);
    });
    acc = hipk_block_sum(acc, sbuf);
    if (threadIdx.x == 0) {
        part_qq[c] = acc;
        if (c == 0) {
            double *rvec = scal->v.rvec;
            for (int j = 0; j <= k; ++j) rvec[j] = ((pass == 0) ? 0.0 : rvec[j]) + hs[j];
        }
    }
}

__device__ __forceinline__ void hipk_givens(double a, double b, double &cs, double &sn) {  // TSL:508-518
    if (fabs(b) == 0.0) {
        cs = 1.0;
        sn = 0.0;
        return;
    }
    if (fabs(a) < fabs(b)) {
        const double t = -(a / b);
        const double r = 1.0 / sqrt(1.0 + fabs(t) * fabs(t));
        cs = r * t;
        sn = r;
    } else {
        const double t = -(b / a);
        const double r = 1.0 / sqrt(1.0 + fabs(t) * fabs(t));
        cs = r;
        sn = r * t;
    }
}

// v_{k+1} = q/||q|| (zero when ||q|| <= eps ||A v_k||), column k of H, breakdown,
// and for 'incremental' the Givens update + early-exit test  (TSL:358-387, 595-623)
template <typename T>
__global__ __launch_bounds__(HIPK_THREADS) void hipk_gm_normalize_kernel(
    int64_t n, int ch, int g, hipk_gm_scal *__restrict__ scal, int k, T *__restrict__ w,
    const double *__restrict__ part_qq, const double *__restrict__ part_ww, double eps, int small_ntiles, int guard = 0) {
    hipk_pre<T, 1> pre;  // w travels while the stop word is read and the partials are folded
    pre.issue(n, ch, blockIdx.x, {(const T *)w});
    if (k >= scal->stop_step) return;
    __shared__ double sbuf[2 * HIPK_THREADS];
    double qq, ww;
    if (small_ntiles > 0) {  // small systems: ||A v||^2 straight from the SpMV's tile sums (no combine launch)
        ww = hipk_fold_tiles8(part_ww, small_ntiles, ch / HIPK_TILE, g, sbuf);
        qq = hipk_reduce_parts(part_qq, g, sbuf);
    } else {
        hipk_reduce_parts2(part_qq, part_ww, g, qq, ww, sbuf);
    }
    if (guard) {
        // SPECULATION: the host did not enqueue this step's second-pass launches (it happens about once per cycle, at the
        // steps the host predicts).  Every workgroup takes the CGS2 decision (TSL:313-326) from the same partials; if the
        // pass was wanted after all, nothing is stored: the step is stopped and reported, the host re-enqueues the cycle
        // from this step with the second pass in place.
        double qnorm;
        if (hipk_gm_want_pass2(scal, k, qq, eps, &qnorm, scal->v.rvec)) {
            if (blockIdx.x == 0 && threadIdx.x == 0) {
                scal->redo = 1;
                scal->redo_step = k;
                scal->stop_step = k;   // this launch's other workgroups compare k >= stop_step too: they return either way
            }
            return;
        }
    }
    double norm1 = sqrt(qq < 0.0 ? 0.0 : qq);
    double norm0 = sqrt(ww < 0.0 ? 0.0 : ww);
    if (!(norm0 > eps)) norm0 = 0.0;
    const double thr = eps * norm0;
    const bool use = norm1 > thr;
    const T nrm = (T)norm1;
    pre.run([&](int64_t i, int nv, T(&v)[1][hipk_vec<T>::VEC]) {
        constexpr int VEC = hipk_vec<T>::VEC;
        T wv[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) wv[e] = use ? v[0][e] / nrm : (T)0;
        hipk_st<T>(w, i, nv, wv);
    });
    // column k of H and the scalar bookkeeping of the step, workgroup 0.  The copies run one element per thread and the Givens
    // chain takes its operands from LDS: as one thread's loop of dependent loads this tail outlasted the kernel's streaming part
    // (the launch then waits for workgroup 0 alone).
    if (blockIdx.x == 0) {
        if (!use) norm1 = 0.0;
        const hipk_gm_view v = scal->v;
        double *H = v.H;
        const int ldh = v.ldh;
        const int t = threadIdx.x;
        __shared__ double hc[HIPK_GM_MAXM_BIG + 2], gvs[2 * (HIPK_GM_MAXM_BIG + 1)];
        if (t <= k) {
            const double rj = v.rvec[t];
            H[t * ldh + k] = rj;
            hc[t] = rj;
        }
        if (scal->incremental)
            for (int i = t; i < 2 * k; i += HIPK_THREADS) gvs[i] = v.gv[i];
        __syncthreads();
        if (t == 0) {
            H[(k + 1) * ldh + k] = norm1;
            hc[k + 1] = norm1;
            scal->steps_done = k + 1;
            bool stop = false;
            if (norm1 == 0.0) {  // TSL:387
                scal->breakdown = 1;
                stop = true;
            }
            if (scal->incremental) {
                for (int i = 0; i < k; ++i) {
                    const double cs = gvs[2 * i], sn = gvs[2 * i + 1];
                    const double p0 = cs * hc[i], p1 = sn * hc[i + 1];
                    const double t0 = p0 - p1;
                    const double p2 = sn * hc[i], p3 = cs * hc[i + 1];
                    hc[i + 1] = p2 + p3;
                    hc[i] = t0;
                }
                double cs, sn;
                hipk_givens(hc[k], hc[k + 1], cs, sn);
                v.gv[2 * k] = cs;
                v.gv[2 * k + 1] = sn;
                {
                    const double p0 = cs * hc[k], p1 = sn * hc[k + 1];
                    hc[k] = p0 - p1;
                }
                hc[k + 1] = 0.0;
                for (int j = 0; j <= k; ++j) v.R[j * ldh + k] = hc[j];
                double *bv = v.beta_vec;
                const double p0 = cs * bv[k], p1 = sn * bv[k + 1];
                const double t0 = p0 - p1;
                const double p2 = sn * bv[k], p3 = cs * bv[k + 1];
                bv[k + 1] = p2 + p3;
                bv[k] = t0;
                const double err = fabs(bv[k + 1]);
                scal->err = err;
                if (!(err > scal->ptol)) stop = true;  // TSL:591
            }
            if (stop) scal->stop_step = k + 1;
        }
    }
}

// Large systems: the normalise step as TWO launches.  hipk_gm_normalize_kernel above makes every workgroup fold the 2 x g chunk
// partials (31 KB from L2 at N = 4 M) before it may scale its 16 KB of w: 64 MB in 16.8 us = 3.8 TB/s, 0.48 of peak, the kernel of
// the GMRES cycle furthest below its roofline.  Here ONE workgroup folds them once, takes the CGS2 guard, thresholds
// the norm and does the H-column / Givens bookkeeping (hipk_gm_hcol_kernel: the same folds, the same bits); a flat grid then
// scales w by the published norm with nothing in front of its loads (hipk_gm_scale_kernel).
__global__ __launch_bounds__(HIPK_THREADS) void hipk_gm_hcol_kernel(int g, hipk_gm_scal *__restrict__ scal, int k,
                                                                    const double *__restrict__ part_qq,
                                                                    const double *__restrict__ part_ww, double eps, int guard,
                                                                    int f32) {
    if (k >= scal->stop_step) return;
    __shared__ double sbuf[2 * HIPK_THREADS];
    double qq, ww;
    hipk_reduce_parts2(part_qq, part_ww, g, qq, ww, sbuf);
    const hipk_gm_view v = scal->v;
    if (guard) {   // speculation miss: see hipk_gm_normalize_kernel
        double qnorm;
        if (hipk_gm_want_pass2(scal, k, qq, eps, &qnorm, v.rvec)) {
            if (threadIdx.x == 0) {
                scal->redo = 1;
                scal->redo_step = k;
                scal->stop_step = k;   // the scale kernel of this step (and everything after it) returns: nothing is stored
            }
            return;
        }
    }
    double norm1 = sqrt(qq < 0.0 ? 0.0 : qq);
    double norm0 = sqrt(ww < 0.0 ? 0.0 : ww);
    if (!(norm0 > eps)) norm0 = 0.0;
    const double thr = eps * norm0;
    const bool use = norm1 > thr;
    double *H = v.H;
    const int ldh = v.ldh;
    const int t = threadIdx.x;
    __shared__ double hc[HIPK_GM_MAXM_BIG + 2], gvs[2 * (HIPK_GM_MAXM_BIG + 1)];
    if (t <= k) {
        const double rj = v.rvec[t];
        H[t * ldh + k] = rj;
        hc[t] = rj;
    }
    if (scal->incremental)
        for (int i = t; i < 2 * k; i += HIPK_THREADS) gvs[i] = v.gv[i];
    __syncthreads();
    if (t == 0) {
        scal->vnorm = f32 ? (double)(float)norm1 : norm1;   // the divisor the scale kernel uses: (T)norm1
        scal->vuse = use ? 1 : 0;
        if (!use) norm1 = 0.0;
        H[(k + 1) * ldh + k] = norm1;
        hc[k + 1] = norm1;
        scal->steps_done = k + 1;
        bool stop = false;
        if (norm1 == 0.0) {  // TSL:387
            scal->breakdown = 1;
            stop = true;
        }
        if (scal->incremental) {
            for (int i = 0; i < k; ++i) {
                const double cs = gvs[2 * i], sn = gvs[2 * i + 1];
                const double p0 = cs * hc[i], p1 = sn * hc[i + 1];
                const double t0 = p0 - p1;
                const double p2 = sn * hc[i], p3 = cs * hc[i + 1];
                hc[i + 1] = p2 + p3;
                hc[i] = t0;
            }
            double cs, sn;
            hipk_givens(hc[k], hc[k + 1], cs, sn);
            v.gv[2 * k] = cs;
            v.gv[2 * k + 1] = sn;
            {
                const double p0 = cs * hc[k], p1 = sn * hc[k + 1];
                hc[k] = p0 - p1;
            }
            hc[k + 1] = 0.0;
            for (int j = 0; j <= k; ++j) v.R[j * ldh + k] = hc[j];
            double *bv = v.beta_vec;
            const double p0 = cs * bv[k], p1 = sn * bv[k + 1];
            const double t0 = p0 - p1;
            const double p2 = sn * bv[k], p3 = cs * bv[k + 1];
            bv[k + 1] = p2 + p3;
            bv[k] = t0;
            const double err = fabs(bv[k + 1]);
            scal->err = err;
            if (!(err > scal->ptol)) stop = true;  // TSL:591
        }
        if (stop) scal->stop_step = k + 1;
    }
}

// v_{k+1} = q / ||q|| (zero when the norm did not pass the threshold): flat grid, 2048 elements per workgroup, w requested before
// the stop word and the norm are read
template <typename T>
__global__ __launch_bounds__(HIPK_THREADS) void hipk_gm_scale_kernel(int64_t n, const hipk_gm_scal *__restrict__ scal, int k,
                                                                     T *__restrict__ w) {
    hipk_pre<T, 1> pre;
    pre.issue(n, HIPK_BASE_CHUNK, blockIdx.x, {(const T *)w});
    if (k >= scal->stop_step) return;
    const T nrm = (T)scal->vnorm;
    const bool use = scal->vuse != 0;
    pre.run([&](int64_t i, int nv, T(&v)[1][hipk_vec<T>::VEC]) {
        constexpr int VEC = hipk_vec<T>::VEC;
        T wv[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) wv[e] = use ? v[0][e] / nrm : (T)0;
        hipk_st<T>(w, i, nv, wv);
    });
}

// =====================================================================================================================
// Small systems (<= 8 reduction chunks of 2048 rows, n <= 16384; rows of <= 32 entries): ONE LAUNCH PER RESTART CYCLE.
// The LDC pressure solve of the reference's example (ldc_solver_module_a.py:17-22: n = 10^4, gmres(restart=30)) is launch
// bound: 6 launches of >= 4.9 us per Arnoldi step for a few hundred KB of traffic.  Here one workgroup PER CHUNK (1024
// threads fp64 / 512 fp32: one 256-thread group per chunk-loop step, the wide form above) stays resident for the whole
// cycle and the workgroups meet at a counter barrier between the phases of a step:
//     A  w = (M) A v_k on the own rows (CSR, thread per row; per-wavefront sums of <w,w>) | multi-dot partials | barrier
//     B  h = fold of the partials | q = w - V h on the own chunk, <q,q> partial                                 | barrier
//     C  CGS2 decision (every workgroup, same bits) [| pass 2: A', B' with two more barriers]
//        | v_{k+1} = q / ||q|| on the own chunk | workgroup 0: H column, Givens, breakdown / early exit           | barrier
// Cross-workgroup data (partials, the new basis vector that the next SpMV gathers, the stop word) is stored write-through
// (sc1) and drained before the arrival, and read with sc1 loads (cdna guide, G16: no fence needed for this hand-off form);
// a workgroup's own chunk of w / V is only ever touched by itself.  All workgroups are placed on ONE XCD (only blocks with
// blockIdx.x % 8 == 0 work: round-robin dispatch; speed only).  Arithmetic = the multi-launch small-system path, bit for bit.
template <typename T>
struct hipk_gm_cyc_args {
    int64_t n;
    int g, m;
    hipk_gm_scal *scal;
    T *V;
    int64_t ldv;
    const int *crow;
    const int *col;
    const T *val;
    const T *dscale;   // left Jacobi preconditioning (HIPK_SPMV_SCALE), or null
    double *part_md;   // [m + 1][HIPK_MAX_PARTS]
    double *part_qq;   // [HIPK_MAX_PARTS]
    double *tile_ww;   // [ntiles * 4] per-wavefront sums of <w,w>
    T *q;              // hipk_gm_solve_lds_kernel: the unnormalised q of the step, gathered by the next SpMV
    int incremental;   //   solve_method 'incremental' (TSL:557-638)
    double ptol;       //   its early-exit threshold (TSL:591)
    double beta0;      //   ||r|| at the start of the cycle (beta_vec[0])
    const T *b;        // hipk_gm_solve_lds_kernel: right-hand side, solution (updated in place),
    T *x;
    double atol_eff;   //   the loop test `res_norm > atol_eff` (TSL:754),
    long long cycles_left;  // cycles the solve may still run (maxiter - cycles so far),
    long long max_cycles;   // and the cycle budget of one launch
    int test_not_resident;  // tests (HIPK_TEST_LDS_NOT_RESIDENT): report the workgroups as not co-resident
    int spread;             // more than 64 workgroups: one per block all over the chip (then LOCAL = false)
    unsigned long long *flag_a, *flag_b;   // [512] each: per-workgroup hand-off flags, zeroed before the launch
    int32_t *bar;      // barrier counter, zeroed by hipk_gm_cycle_init_kernel
    double eps;
    unsigned long long *stamps;  // diagnostic (HIPK_GM_STAMPS=1): per-phase shader-clock totals of workgroup 0, else null
};

// hipk_fold8 on partials another workgroup of THIS launch wrote
__device__ __forceinline__ double hipk_fold8_sc1(const double *part, int g) {
    double a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = 0.0 + ((i < g) ? hipk_peek(part + i) : 0.0);
    return ((a[0] + a[4]) + (a[2] + a[6])) + ((a[1] + a[5]) + (a[3] + a[7]));
}
// hipk_wave_fold / hipk_fold_tiles8 with sc1 loads; any number of wavefronts may call (all threads must)
__device__ __forceinline__ double hipk_wave_fold_sc1(const double *tp, int cnt, int lane) {
    double a[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int t = lane + 64 * j;
        double acc = 0.0;
        for (int i = t; i < cnt; i += HIPK_THREADS) {
            const double *w4 = tp + (size_t)i * 4;
            const double w0 = hipk_peek(w4), w1 = hipk_peek(w4 + 1), w2 = hipk_peek(w4 + 2), w3 = hipk_peek(w4 + 3);
            acc = acc + ((w0 + w1) + (w2 + w3));
        }
        a[j] = acc;
    }
    a[0] = a[0] + a[2];
    a[1] = a[1] + a[3];
    double v = a[0] + a[1];
    v = hipk_wave_sum(v);
    return v;
}
__device__ __forceinline__ double hipk_fold_tiles8_sc1(const double *tp, int ntiles, int tpc, int g, double *cp, int nwaves) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int c = wave; c < 8; c += nwaves) {
        double r = 0.0;
        if (c < g) {
            const int first = c * tpc;
            const int cnt = (ntiles - first < tpc) ? ntiles - first : tpc;
            r = hipk_wave_fold_sc1(tp + (size_t)first * 4, cnt, lane);
        }
        if (lane == 0) cp[c] = r;
    }
    __syncthreads();
    double a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = 0.0 + ((i < g) ? cp[i] : 0.0);
    const double v = ((a[0] + a[4]) + (a[2] + a[6])) + ((a[1] + a[5]) + (a[3] + a[7]));
    __syncthreads();
    return v;
}
template <typename T>
__global__ __launch_bounds__(HIPK_BASE_CHUNK / hipk_vec<T>::VEC) void hipk_gm_cycle_small_kernel(hipk_gm_cyc_args<T> a) {
    constexpr int VEC = hipk_vec<T>::VEC;
    constexpr int NTH = HIPK_BASE_CHUNK / VEC;       // threads per workgroup
    constexpr int NIT = NTH / HIPK_THREADS;          // 256-thread groups = chunk-loop steps
    constexpr int NW = NTH / 64;
    if (blockIdx.x & 7) return;                      // the working blocks share an XCD (dispatch is round-robin over 8)
    const int c = blockIdx.x >> 3;
    const int g = a.g;
    if (c >= g) return;
    hipk_gm_scal *scal = a.scal;
    const int tid = threadIdx.x, lane = tid & 63;
    const int t = tid & (HIPK_THREADS - 1), q = tid / HIPK_THREADS;
    const int64_t n = a.n;
    const int64_t base = (int64_t)c * HIPK_BASE_CHUNK;
    const int64_t end = (base + HIPK_BASE_CHUNK < n) ? base + HIPK_BASE_CHUNK : n;
    const int64_t ei = base + (int64_t)VEC * t + (int64_t)q * VEC * HIPK_THREADS;   // this thread's element group
    const int nv = (ei < end) ? ((end - ei < VEC) ? (int)(end - ei) : VEC) : 0;
    const int ntiles = (int)((n + HIPK_TILE - 1) / HIPK_TILE);
    const int *__restrict__ crow = a.crow;
    const int *__restrict__ col = a.col;
    const T *__restrict__ val = a.val;

    __shared__ double chain[8 * HIPK_THREADS];
    __shared__ double hs[HIPK_GM_LDH];
    __shared__ double rv[HIPK_GM_LDH];
    __shared__ double cp[8];
    __shared__ double bc[2];
    __shared__ double hc[HIPK_GM_LDH + 1];   // the Hessenberg column under rotation (thread 0 of workgroup 0)
    __shared__ int fail;
    __shared__ long long stop_lds;
    if (tid == 0) fail = 0;
    if (tid < HIPK_GM_LDH) rv[tid] = 0.0;
    __syncthreads();
    int epoch = 0;
    long long stop = a.m;
    // phase stamps: a DIAGNOSTIC build only (make STAMPS=1 -> -DHIPK_GM_STAMPS; run with HIPK_GM_STAMPS=1)
#ifdef HIPK_GM_STAMPS
    unsigned long long t_prev = 0, t_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const bool stamping = a.stamps != nullptr && c == 0 && tid == 0;
#define HIPK_STAMP(slot)                                              \
    if (stamping) {                                                   \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
        t_acc[slot] += now_ - t_prev;                                 \
        t_prev = now_;                                                \
    }
    if (stamping) t_prev = __builtin_amdgcn_s_memtime();
#else
#define HIPK_STAMP(slot)
#endif

    for (int k = 0; k < a.m && k < stop; ++k) {
        const T *vk = a.V + (int64_t)k * a.ldv;
        T *w = a.V + (int64_t)(k + 1) * a.ldv;
        // ---------------- A: w = (M) A v_k on the own rows, per-wavefront sums of <w,w>
#pragma unroll
        for (int i = 0; i < HIPK_BASE_CHUNK / NTH; ++i) {
            const int64_t r = base + (int64_t)i * NTH + tid;
            T out = (T)0;
            if (r < n) {
                const int lo = crow[r], hi = crow[r + 1];
                T s = (T)0;
                for (int j = lo; j < hi; ++j) {
                    const T p = val[j] * hipk_peek_t<T>(vk + col[j]);
                    s = s + p;
                }
                out = s;
                if (a.dscale) out = a.dscale[r] * out;
                w[r] = out;
            }
            double d1 = (r < n) ? (double)out * (double)out : 0.0;
            d1 = hipk_wave_sum(d1);
            const int64_t rf = r - lane;   // first row of this wavefront's 64
            if (lane == 0 && rf / HIPK_TILE < ntiles) hipk_publish(&a.tile_ww[(size_t)(rf / HIPK_TILE) * 4 + (rf % HIPK_TILE) / 64], d1);
        }
        __syncthreads();   // w is read below by other threads of this workgroup
        HIPK_STAMP(0)
        for (int pass = 0; pass < 2; ++pass) {
            // ---------------- multi-dot partials of the own chunk (wide form: group qq owns chunk-loop step qq)
            T wv[VEC];
            if (nv > 0) hipk_ld<T>((const T *)w, ei, nv, wv);
            for (int j0 = 0; j0 <= k; j0 += 8) {
                T vv[8][VEC];
                if (nv > 0) {
#pragma unroll
                    for (int b = 0; b < 8; ++b)
                        if (j0 + b <= k) hipk_ld<T>(a.V + (int64_t)(j0 + b) * a.ldv, ei, nv, vv[b]);
                }
#pragma unroll
                for (int qq = 0; qq < NIT; ++qq) {
                    if (q == qq) {
#pragma unroll
                        for (int b = 0; b < 8; ++b) {
                            double acc = (qq == 0) ? 0.0 : chain[b * HIPK_THREADS + t];
                            if (j0 + b <= k) {
#pragma unroll
                                for (int e = 0; e < VEC; ++e)
                                    if (e < nv) acc = fma((double)vv[b][e], (double)wv[e], acc);
                            }
                            chain[b * HIPK_THREADS + t] = acc;
                        }
                    }
                    __syncthreads();
                }
                if (tid < 128) {
#pragma unroll
                    for (int b = 0; b < 8; ++b) chain[b * HIPK_THREADS + tid] = chain[b * HIPK_THREADS + tid] + chain[b * HIPK_THREADS + tid + 128];
                }
                __syncthreads();
                if (tid < 64) {
#pragma unroll
                    for (int b = 0; b < 8; ++b) {
                        double v = chain[b * HIPK_THREADS + tid] + chain[b * HIPK_THREADS + tid + 64];
                        v = hipk_wave_sum(v);
                        if (tid == 0 && j0 + b <= k) hipk_publish(&a.part_md[(size_t)(j0 + b) * HIPK_MAX_PARTS + c], v);
                    }
                }
                __syncthreads();
            }
            HIPK_STAMP(1)
            if (!hipk_gbar(a.bar, g, epoch, &fail)) {
                if (tid == 0) scal->redo = -1;
                return;
            }
            HIPK_STAMP(2)
            // ---------------- B: h = fold of the partials; q = w - V h on the own chunk; <q,q> partial
            if (tid < HIPK_GM_LDH) {
                const double hj = (tid <= k) ? hipk_fold8_sc1(a.part_md + (size_t)tid * HIPK_MAX_PARTS, g) : 0.0;
                hs[tid] = hj;
                rv[tid] = ((pass == 0) ? 0.0 : rv[tid]) + hj;   // rvec += h (TSL:305), kept by every workgroup
            }
            __syncthreads();
            if (nv > 0) {
                double sacc[VEC];
#pragma unroll
                for (int e = 0; e < VEC; ++e) sacc[e] = 0.0;
                for (int j0 = 0; j0 <= k; j0 += 8) {
                    T vv[8][VEC];
#pragma unroll
                    for (int b = 0; b < 8; ++b)
                        if (j0 + b <= k) hipk_ld<T>(a.V + (int64_t)(j0 + b) * a.ldv, ei, nv, vv[b]);
#pragma unroll
                    for (int b = 0; b < 8; ++b)
                        if (j0 + b <= k) {
                            const double hj = hs[j0 + b];
#pragma unroll
                            for (int e = 0; e < VEC; ++e) sacc[e] = fma((double)vv[b][e], hj, sacc[e]);
                        }
                }
#pragma unroll
                for (int e = 0; e < VEC; ++e) wv[e] = (T)((double)wv[e] - sacc[e]);
                hipk_st<T>(w, ei, nv, wv);
            }
#pragma unroll
            for (int qq = 0; qq < NIT; ++qq) {   // the spec's per-thread <q,q> chain, one chunk-loop step per group
                if (q == qq) {
                    double acc = (qq == 0) ? 0.0 : chain[t];
#pragma unroll
                    for (int e = 0; e < VEC; ++e)
                        if (e < nv) acc = fma((double)wv[e], (double)wv[e], acc);
                    chain[t] = acc;
                }
                __syncthreads();
            }
            if (tid < 128) chain[tid] = chain[tid] + chain[tid + 128];
            __syncthreads();
            if (tid < 64) {
                double v = chain[tid] + chain[tid + 64];
                v = hipk_wave_sum(v);
                if (tid == 0) hipk_publish(&a.part_qq[c], v);
            }
            HIPK_STAMP(3)
            if (!hipk_gbar(a.bar, g, epoch, &fail)) {
                if (tid == 0) scal->redo = -1;
                return;
            }
            HIPK_STAMP(4)
            // ---------------- C: CGS2 decision (TSL:313-326), every workgroup from the same partials
            if (pass == 1) break;
            if (tid == 0) {
                const double qq = hipk_fold8_sc1(a.part_qq, g);
                double qnorm = sqrt(qq < 0.0 ? 0.0 : qq);
                if (!(qnorm > a.eps)) qnorm = 0.0;
                double rr = 0.0;
                for (int j = 0; j <= k; ++j) rr = fma(rv[j], rv[j], rr);
                double rnorm = sqrt(rr < 0.0 ? 0.0 : rr);
                if (!(rnorm > a.eps)) rnorm = 0.0;
                bc[0] = (rnorm < qnorm * HIPK_INV_SQRT2) ? 1.0 : 0.0;
            }
            __syncthreads();
            const bool want2 = bc[0] != 0.0;
            __syncthreads();
            if (!want2) break;
        }
        // ---------------- normalise: v_{k+1} = q / ||q|| (zero when ||q|| <= eps ||A v_k||), TSL:358-387
        const double ww = hipk_fold_tiles8_sc1(a.tile_ww, ntiles, HIPK_BASE_CHUNK / HIPK_TILE, g, cp, NW);
        if (tid == 0) bc[1] = hipk_fold8_sc1(a.part_qq, g);
        __syncthreads();
        const double qq = bc[1];
        double norm1 = sqrt(qq < 0.0 ? 0.0 : qq);
        double norm0 = sqrt(ww < 0.0 ? 0.0 : ww);
        if (!(norm0 > a.eps)) norm0 = 0.0;
        const double thr = a.eps * norm0;
        const bool use = norm1 > thr;
        const T nrm = (T)norm1;
        if (nv > 0) {
            T qv[VEC];
            hipk_ld<T>((const T *)w, ei, nv, qv);
#pragma unroll
            for (int e = 0; e < VEC; ++e)
                if (e < nv) hipk_publish_t(w + ei + e, use ? qv[e] / nrm : (T)0);   // gathered by the other workgroups' next SpMV
        }
        if (c == 0 && tid == 0) {
            if (!use) norm1 = 0.0;
            double *H = scal->H;
            for (int j = 0; j <= k; ++j) H[j * HIPK_GM_LDH + k] = rv[j];
            H[(k + 1) * HIPK_GM_LDH + k] = norm1;
            scal->steps_done = k + 1;
            bool stp = false;
            if (norm1 == 0.0) {  // TSL:387
                scal->breakdown = 1;
                stp = true;
            }
            if (scal->incremental) {
                for (int j = 0; j <= k + 1; ++j) hc[j] = H[j * HIPK_GM_LDH + k];
                for (int i = 0; i < k; ++i) {
                    const double cs = scal->gv[2 * i], sn = scal->gv[2 * i + 1];
                    const double p0 = cs * hc[i], p1 = sn * hc[i + 1];
                    const double t0 = p0 - p1;
                    const double p2 = sn * hc[i], p3 = cs * hc[i + 1];
                    hc[i + 1] = p2 + p3;
                    hc[i] = t0;
                }
                double cs, sn;
                hipk_givens(hc[k], hc[k + 1], cs, sn);
                scal->gv[2 * k] = cs;
                scal->gv[2 * k + 1] = sn;
                {
                    const double p0 = cs * hc[k], p1 = sn * hc[k + 1];
                    hc[k] = p0 - p1;
                }
                hc[k + 1] = 0.0;
                for (int j = 0; j <= k; ++j) scal->R[j * HIPK_GM_LDH + k] = hc[j];
                double *bv = scal->beta_vec;
                const double p0 = cs * bv[k], p1 = sn * bv[k + 1];
                const double t0 = p0 - p1;
                const double p2 = sn * bv[k], p3 = cs * bv[k + 1];
                bv[k + 1] = p2 + p3;
                bv[k] = t0;
                const double err = fabs(bv[k + 1]);
                scal->err = err;
                if (!(err > scal->ptol)) stp = true;  // TSL:591
            }
            if (stp) __hip_atomic_store((long long *)&scal->stop_step, (long long)(k + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        HIPK_STAMP(5)
        if (!hipk_gbar(a.bar, g, epoch, &fail)) {
            if (tid == 0) scal->redo = -1;
            return;
        }
        HIPK_STAMP(6)
        if (tid == 0) stop_lds = __hip_atomic_load((const long long *)&scal->stop_step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        stop = stop_lds;
        __syncthreads();
    }
#ifdef HIPK_GM_STAMPS
    if (stamping)
        for (int i = 0; i < 8; ++i) a.stamps[i] += t_acc[i];
#endif
#undef HIPK_STAMP
}

// =====================================================================================================================
// The same cycle with the BASIS RESIDENT IN LDS and each reduction chunk shared by EIGHT workgroups.
// hipk_gm_cycle_small_kernel keeps one workgroup per chunk busy: at n = 10^4 that is 5 compute units, and every multi-dot /
// update re-reads the live columns of its chunk from L2 (a step costs ~33 us).  The spec's chunk tree
// (v[t] += v[t+s], s = 128 .. 1 over the 256 virtual threads) keeps the residues t mod 8 apart until its last three levels:
// sub-workgroup s of chunk c owns the 32 virtual threads t = s + 8u (their 256 rows: 16-byte pieces, 128 bytes apart), runs
// their chains and the five tree levels s = 128 .. 8 on its own (a 32-lane register reduction) and publishes a SUB-partial;
// whoever needs the chunk partial adds the eight sub-partials ((p0+p4)+(p2+p6))+((p1+p5)+(p3+p7)) -- the tree's levels
// s = 4, 2, 1.  Same additions, same order, same bits as the one-workgroup-per-chunk form.
//   * thread (u = tid & 31, e = tid >> 5) owns ONE row -- element e (0..7) of virtual thread u's chain -- with the row's
//     matrix entries in registers for the whole cycle (rows > 16 entries: the tail is re-read);
//   * the 256 rows x m basis columns of the sub-workgroup live in LDS (61 KB fp64 at restart 30): multi-dot and update
//     never touch memory; the only global traffic of a step is the new vector (published for the gathers of the next SpMV
//     and for the x update after the cycle), the sub-partials and the tile sums of <w,w>;
//   * <w,w> is the TILED dot of the SpMV epilogue (wavefront sums over 64 CONTIGUOUS rows): sub-workgroup s re-reads tile s
//     of its chunk (published by the eight owners of those rows) one barrier later and forms the four wavefront sums.
// Three counter barriers per step as before; 8 g workgroups (<= 64) on one XCD, at most two per compute unit.
// column k of H from rvec and ||q||, breakdown, and for 'incremental' the Givens update + early-exit test (TSL:358-387,
// 595-623); one thread.  Returns true when the cycle stops after this step.
__device__ __forceinline__ bool hipk_gm_hcolumn(hipk_gm_scal *scal, int k, const double *rv, double norm1, double *hc) {
    double *H = scal->H;
    for (int j = 0; j <= k; ++j) H[j * HIPK_GM_LDH + k] = rv[j];
    H[(k + 1) * HIPK_GM_LDH + k] = norm1;
    scal->steps_done = k + 1;
    bool stp = false;
    if (norm1 == 0.0) {  // TSL:387
        scal->breakdown = 1;
        stp = true;
    }
    if (scal->incremental) {
        for (int j = 0; j <= k + 1; ++j) hc[j] = H[j * HIPK_GM_LDH + k];
        for (int i = 0; i < k; ++i) {
            const double cs = scal->gv[2 * i], sn = scal->gv[2 * i + 1];
            const double p0 = cs * hc[i], p1 = sn * hc[i + 1];
            const double t0 = p0 - p1;
            const double p2 = sn * hc[i], p3 = cs * hc[i + 1];
            hc[i + 1] = p2 + p3;
            hc[i] = t0;
        }
        double cs, sn;
        hipk_givens(hc[k], hc[k + 1], cs, sn);
        scal->gv[2 * k] = cs;
        scal->gv[2 * k + 1] = sn;
        {
            const double p0 = cs * hc[k], p1 = sn * hc[k + 1];
            hc[k] = p0 - p1;
        }
        hc[k + 1] = 0.0;
        for (int j = 0; j <= k; ++j) scal->R[j * HIPK_GM_LDH + k] = hc[j];
        double *bv = scal->beta_vec;
        const double p0 = cs * bv[k], p1 = sn * bv[k + 1];
        const double t0 = p0 - p1;
        const double p2 = sn * bv[k], p3 = cs * bv[k + 1];
        bv[k + 1] = p2 + p3;
        bv[k] = t0;
        const double err = fabs(bv[k + 1]);
        scal->err = err;
        if (!(err > scal->ptol)) stp = true;  // TSL:591
    }
    return stp;
}

static constexpr int kGmRowRegs = 12;   // matrix entries of the own row held in registers
// LDS carve of hipk_gm_solve_lds_kernel (doubles after the T arrays)
struct hipk_gm_lds_off {
    static constexpr int hs = 0;                                   // [32]
    static constexpr int rv = hs + HIPK_GM_LDH;                    // [32]
    static constexpr int bc = rv + HIPK_GM_LDH;                    // [8]  broadcast slots
    static constexpr int gv = bc + 8;                              // [64] Givens rotations of the cycle
    static constexpr int bv = gv + 2 * HIPK_GM_LDH;                // [34] beta_vec of the cycle
    static constexpr int hr = bv + HIPK_GM_LDH + 2;                // [33][32] H ('batched') or R ('incremental') of the cycle
    static constexpr int lp = hr + (HIPK_GM_LDH + 1) * HIPK_GM_LDH;  // [496] packed lower triangle: A^T A, then its Cholesky factor
    static constexpr int yl = lp + HIPK_GM_LDH * (HIPK_GM_LDH - 1) / 2;  // [32] y
    static constexpr int zl = yl + HIPK_GM_LDH;                    // [32] z
    static constexpr int total = zl + HIPK_GM_LDH;
};
template <typename T>
static inline size_t hipk_gm_solve_lds_bytes(int m) {
    return (size_t)(m + 1) * 256 * sizeof(T) + (size_t)hipk_gm_lds_off::total * sizeof(double) + 64;
}

// ---- the small dense solves of a cycle, in REGISTERS of wavefront 0 (lane i owns row i)
__device__ __forceinline__ double hipk_readlane_d(double v, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
// 'batched' (TSL:391-421): Cholesky of the normal equations (lower triangle packed in LDS at O::lp, right-hand side at O::zl),
// forward and back substitution, y -> O::yl.  Right-looking in registers: after column p is known every entry (i, j) of the
// trailing triangle takes its p-th chain link fma(-L[i][p], L[j][p], .) -- for each entry the links still come in ascending p,
// the order of hipk_lstsq_normal / the oracle.  Fully unrolled (compile-time register indices; uniform branches skip what lies
// beyond k): 15 k instructions of straight-line code, of which a k = 30 solve executes ~5 k.  A rolled variant (rows shifted
// by one register per step) executed more instructions and was slower (0.321 vs 0.308 ms per cycle); keeping the triangle in
// LDS cost 0.333.  Sets bc[4] = 0 when a pivot is not positive.  'incremental' (TSL:630): the triangular system
// R y = beta_vec from the LDS copies.  Called by wavefront 0 only.
__device__ __noinline__ void hipk_gm_small_solve_wave0(double *sm, int k, int incremental) {
    using O = hipk_gm_lds_off;
    constexpr int KM = HIPK_GM_MAXM;
    k = __builtin_amdgcn_readfirstlane(k);                     // arguments of a device function arrive in vector registers:
    incremental = __builtin_amdgcn_readfirstlane(incremental);  // make the branch conditions scalar again
    double *Lp = sm + O::lp, *yl = sm + O::yl, *zl = sm + O::zl, *bc = sm + O::bc;
    const int lane = threadIdx.x & 63;
    const int ri = lane * (lane + 1) / 2;
    const bool mine = lane < k;
    double c[KM];   // back substitution: c[p] multiplies y_p, p > lane
    double zi, di;
    if (incremental) {
        const double *Rl = sm + O::hr, *bv = sm + O::bv;
#pragma unroll
        for (int p = 0; p < KM; ++p) c[p] = (mine && p > lane && p < k) ? Rl[lane * HIPK_GM_LDH + p] : 0.0;
        zi = mine ? bv[lane] : 0.0;
        di = mine ? Rl[lane * HIPK_GM_LDH + lane] : 1.0;
    } else {
        double r[KM];   // row `lane` of A^T A, becoming row `lane` of L
#pragma unroll
        for (int j = 0; j < KM; ++j) r[j] = (mine && j <= lane) ? Lp[ri + j] : 0.0;
        double si = mine ? zl[lane] : 0.0;   // b2, then the forward substitution's running sums
        zi = 0.0;
        bool ok = true;
#pragma unroll
        for (int p = 0; p < KM; ++p) {
            if (ok && p < k) {   // uniform
                const double d = hipk_readlane_d(r[p], p);   // entry (p, p), all its links applied
                if (!(d > 0.0)) {
                    ok = false;
                } else {
                    const double lpp = sqrt(d);
                    const double lip = (lane == p) ? lpp : r[p] / lpp;   // L[lane][p] (lanes < p: unused)
                    r[p] = lip;
                    // forward: z_p = s_p / L[p][p]; lanes i > p take the link fma(-L[i][p], z_p, s_i)
                    const double zp = hipk_readlane_d(si / lpp, p);
                    zi = (lane == p) ? zp : zi;
                    si = fma(-lip, zp, si);
#pragma unroll
                    for (int j = p + 1; j < KM; ++j)
                        if (j < k) r[j] = fma(-lip, hipk_readlane_d(lip, j), r[j]);   // lanes < j: an entry above the diagonal, unused
                }
            }
        }
        if (!ok) {
            if (lane == 0) bc[4] = 0.0;
            return;
        }
        // transpose through LDS: lane i needs column i of L for the back substitution
#pragma unroll
        for (int j = 0; j < KM; ++j)
            if (mine && j <= lane) Lp[ri + j] = r[j];
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int p = 0; p < KM; ++p) c[p] = (mine && p > lane && p < k) ? Lp[p * (p + 1) / 2 + lane] : 0.0;
        di = mine ? Lp[ri + lane] : 1.0;
    }
    // y_i = (z_i - sum_{p > i} c_i[p] y_p) / d_i, the chain in ascending p (TSL:630 and the back substitution of TSL:418): one
    // lane at a time; a finished y_p is broadcast once
    double yi = 0.0;
    double yu[KM];
#pragma unroll
    for (int i = KM - 1; i >= 0; --i) {
        if (i < k) {   // uniform
            double sacc = zi;
#pragma unroll
            for (int p = i + 1; p < KM; ++p)
                if (p < k) sacc = fma(-c[p], yu[p], sacc);   // meaningful in lane i only
            const double q = sacc / di;
            yi = (lane == i) ? q : yi;
            yu[i] = hipk_readlane_d(q, i);
        }
    }
    if (mine) yl[lane] = yi;
}

// the lower triangle of H^T H (each entry its own fma chain over p = 0..k, the order of hipk_lstsq_normal) and b2 = H[0][:] beta0
// into LDS, by all threads; then hipk_gm_small_solve_wave0.  Returns false when the factorisation met a non-positive pivot.
__device__ __forceinline__ bool hipk_gm_lstsq_lds(double *sm, int k, double beta0, int tid) {
    using O = hipk_gm_lds_off;
    const double *Hl = sm + O::hr;
    double *Lp = sm + O::lp, *zl = sm + O::zl, *bc = sm + O::bc;
    const int nent = k * (k + 1) / 2;
    for (int idx = tid; idx < nent; idx += HIPK_THREADS) {
        int i = (int)((sqrt(8.0 * idx + 1.0) - 1.0) * 0.5);
        while (i * (i + 1) / 2 > idx) --i;
        while ((i + 1) * (i + 2) / 2 <= idx) ++i;
        const int j = idx - i * (i + 1) / 2;
        double acc = 0.0;
        for (int p0 = 0; p0 <= k; p0 += 8) {
            double a8[8], b8[8];
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const int pp = (p0 + b <= k) ? p0 + b : k;
                a8[b] = Hl[pp * HIPK_GM_LDH + i];
                b8[b] = Hl[pp * HIPK_GM_LDH + j];
            }
#pragma unroll
            for (int b = 0; b < 8; ++b)
                if (p0 + b <= k) acc = fma(a8[b], b8[b], acc);
        }
        Lp[idx] = acc;
    }
    if (tid < k) zl[tid] = Hl[tid] * beta0;   // b2[i] = H[0][i] * beta0
    if (tid == 0) bc[4] = 1.0;
    __syncthreads();
    if (tid < 64) hipk_gm_small_solve_wave0(sm, k, 0);
    __syncthreads();
    return bc[4] != 0.0;
}
__device__ __forceinline__ void hipk_gm_trisolve_lds(double *sm, int k, int tid) {
    if (tid < 64) hipk_gm_small_solve_wave0(sm, k, 1);
    __syncthreads();
}

// The WHOLE restarted solve of a small system in one launch: restart cycles one after the other -- Arnoldi steps as described
// above, then on the device what the host path does between two cycle launches: y (every workgroup solves the small
// least-squares problem itself, from its own LDS copy of H or R), x += V y on the own rows from the LDS basis, the residual
// b - A x with its tiled norm, the unit residual as column 0, and the loop test (TSL:754-764).  The host reads one small report
// per launch (a launch is bounded to `max_cycles` cycles).
template <typename T, bool LOCAL>
__global__ __launch_bounds__(HIPK_THREADS, 2) void hipk_gm_solve_lds_kernel(hipk_gm_cyc_args<T> a) {
    using O = hipk_gm_lds_off;
    constexpr int VEC = hipk_vec<T>::VEC;
    int wg = blockIdx.x;                             // spread (more than 64 workgroups): one per block, anywhere on the chip
    if (!a.spread) {
        if (blockIdx.x & 7) return;                  // the working blocks share an XCD (dispatch is round-robin over 8)
        wg = blockIdx.x >> 3;
    }
    const int c = wg / kGmSub, s = wg % kGmSub;
    const int g = a.g, nwg = g * kGmSub;
    if (c >= g) return;
    hipk_gm_scal *scal = a.scal;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int u = tid & 31, e8 = tid >> 5;           // virtual thread s + 8u of the chunk, element e8 of its chain
    const int64_t n = a.n;
    const int m = a.m;
    const int64_t base = (int64_t)c * HIPK_BASE_CHUNK;
    const int64_t row = base + (int64_t)VEC * (s + kGmSub * u) + (int64_t)(e8 / VEC) * (VEC * HIPK_THREADS) + (e8 % VEC);
    const bool live = row < n;
    const int ntiles = (int)((n + HIPK_TILE - 1) / HIPK_TILE);
    const int tile = c * (HIPK_BASE_CHUNK / HIPK_TILE) + s;          // the tile whose wavefront sums this workgroup forms
    const int64_t trow = (int64_t)tile * HIPK_TILE + tid;

    extern __shared__ __align__(16) unsigned char hipk_gm_lds_raw[];
    T *Vl = (T *)hipk_gm_lds_raw;                                    // [m][8][32]
    T *wl = Vl + (size_t)m * 256;                                    // [8][32]
    double *sm = (double *)(hipk_gm_lds_raw + (size_t)(m + 1) * 256 * sizeof(T));   // a multiple of 1 KB
    double *hs = sm + O::hs, *rv = sm + O::rv, *bc = sm + O::bc, *gvl = sm + O::gv, *bvl = sm + O::bv, *HRl = sm + O::hr;
    double *yl = sm + O::yl;
    int *fail = (int *)(sm + O::total);
    unsigned long long *res_lds = (unsigned long long *)(sm + O::total + 1);
    if (tid == 0) *fail = 0;
    for (int i = tid; i < (HIPK_GM_LDH + 1) * HIPK_GM_LDH; i += HIPK_THREADS) HRl[i] = 0.0;   // H below its subdiagonal stays zero

    // the own row of the matrix, of b and of x, for the whole solve
    int lo = 0, len = 0;
    if (live) {
        lo = a.crow[row];
        len = a.crow[row + 1] - lo;
    }
    unsigned cj[kGmRowRegs];   // byte offsets of the row's columns
    T vj[kGmRowRegs];
#pragma unroll
    for (int j = 0; j < kGmRowRegs; ++j) {
        cj[j] = (j < len) ? (unsigned)a.col[lo + j] * (unsigned)sizeof(T) : 0u;
        vj[j] = (j < len) ? a.val[lo + j] : (T)0;
    }
    const T dsc = (a.dscale && live) ? a.dscale[row] : (T)1;
    const T b_own = live ? a.b[row] : (T)0;
    T x_own = live ? a.x[row] : (T)0;
    int wmax = len < kGmRowRegs ? len : kGmRowRegs;   // register-held entries of the longest row of this wavefront
    for (int off = 32; off > 0; off >>= 1) {
        const int o = __shfl_xor(wmax, off);
        wmax = o > wmax ? o : wmax;
    }
    wmax = __builtin_amdgcn_readfirstlane(wmax);
    Vl[tid] = live ? a.V[row] : (T)0;                // column 0: written by the launch before this one
    int epoch = 0;
    // every workgroup resident -- and, LOCAL, all on one XCD?  (the only exchange of this launch that does not rely on the
    // placement: agent-scope atomics.)  Nothing has been modified yet: a failure leaves the solve to the launch sequences.
    if (LOCAL && tid == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        __hip_atomic_fetch_or(&scal->xcc_mask, 1u << (xcc & 15u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!hipk_gbar(a.bar, nwg, epoch, fail) || a.test_not_resident) {
        if (tid == 0) scal->redo = -1;
        return;
    }
    if (LOCAL) {
        if (tid == 0) {
            const unsigned mask = __hip_atomic_load(&scal->xcc_mask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *fail = (__builtin_popcount(mask) == 1) ? 0 : 1;
        }
        __syncthreads();
        if (*fail) {                                 // uniform over the launch: every workgroup reads the same mask
            if (tid == 0) scal->redo = -2;
            return;
        }
    }
    unsigned long long seq = 0;                      // hand-offs so far (uniform over the launch)
    // the vector the next SpMV gathers: column 0 as the launch before left it (first cycle), afterwards the UNNORMALISED q or
    // residual in a.q, divided by its norm on the fly (same division, same bits as the owner's own copy)
    bool from_q = false;
    T nrm_prev = (T)1;
    bool use_prev = true;
    double beta0 = a.beta0;                          // ||r|| at the start of the cycle
    long long cycles = 0, matvecs = 0;
    int status = 0, breakdown_any = 0;
#define HIPK_HO(flags)                                                         \
    if (hipk_ho_sync<LOCAL>(flags, wg, nwg, ++seq, 0u, res_lds) == ~0ull) {    \
        if (tid == 0) scal->redo = -1;                                         \
        return;                                                                \
    }
#ifdef HIPK_GM_STAMPS
    unsigned long long t_prev = 0, t_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const bool stamping = a.stamps != nullptr && wg == 0 && tid == 0;
#define HIPK_STAMP(slot)                                              \
    if (stamping) {                                                   \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
        t_acc[slot] += now_ - t_prev;                                 \
        t_prev = now_;                                                \
    }
    if (stamping) t_prev = __builtin_amdgcn_s_memtime();
#else
#define HIPK_STAMP(slot)
#endif

    for (;;) {   // restart cycles
        if (tid < HIPK_GM_LDH) rv[tid] = 0.0;
        if (tid == 0) {
            bvl[0] = beta0;
            bvl[1] = 0.0;
            bc[5] = 0.0;
            bc[6] = 0.0;
            if (wg == 0) scal->breakdown = 0;
        }
        int kk = m;                                  // Arnoldi steps of this cycle
        for (int k = 0; k < m; ++k) {
            // ---------------- A: w = (M) A v_k on the own row
            const T *src = from_q ? a.q : a.V;
            T *wcol = a.V + (int64_t)(k + 1) * a.ldv;
            T xs[kGmRowRegs];
#pragma unroll
            for (int j = 0; j < kGmRowRegs; ++j)
                if (j < wmax) xs[j] = hipk_peek_off<T>(src, cj[j]);   // uniform per wavefront; unused slots gather entry 0
            T acc_row = (T)0;
#pragma unroll
            for (int j = 0; j < kGmRowRegs; ++j)
                if (j < wmax) {
                    const T xv = !from_q ? xs[j] : (use_prev ? xs[j] / nrm_prev : (T)0);
                    const T p = vj[j] * xv;
                    acc_row = (j < len) ? acc_row + p : acc_row;
                }
            for (int j = kGmRowRegs; j < len; ++j) {
                T xv = hipk_peek_t<T>(src + a.col[lo + j]);
                if (from_q) xv = use_prev ? xv / nrm_prev : (T)0;
                const T p = a.val[lo + j] * xv;
                acc_row = acc_row + p;
            }
            T w_own = acc_row;
            if (a.dscale) w_own = dsc * w_own;
            if (!live) w_own = (T)0;
            wl[tid] = w_own;
            if (live) hipk_ho_store<LOCAL>(wcol + row, w_own);   // for the tile sums of <w,w> (formed by the tile's workgroup in B)
            __syncthreads();
            HIPK_STAMP(0)
            for (int pass = 0; pass < 2; ++pass) {
                // ---------------- multi-dot sub-partials: thread (u, e8) runs the chains of columns e8, e8 + 8, e8 + 16, e8 + 24
                // (all LDS loads of a column group issued before its chain: no per-element branches)
                {
                    double wv[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) wv[e] = (double)wl[e * 32 + u];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (8 * i <= k) {   // uniform
                            const int j = e8 + 8 * i;
                            const bool on = j <= k;
                            const int jj = on ? j : 0;
                            T vv[8];
#pragma unroll
                            for (int e = 0; e < 8; ++e) vv[e] = Vl[((size_t)jj * 8 + e) * 32 + u];
                            double acc = 0.0;
#pragma unroll
                            for (int e = 0; e < 8; ++e) acc = fma((double)vv[e], wv[e], acc);
                            const double v = hipk_half_sum(acc);
                            if (u == 0 && on) hipk_ho_store<LOCAL>(&a.part_md[(size_t)j * HIPK_MAX_PARTS + wg], v);
                        }
                    }
                }
                HIPK_STAMP(1)
                HIPK_HO(a.flag_a)
                HIPK_STAMP(2)
                // ---------------- B: h = fold of the sub-partials (thread (j, chunk i8): 8 loads, register folds)
                T wt = (T)0;
                if (pass == 0 && trow < n) wt = hipk_peek_t<T>(wcol + trow);     // tile sums of <w,w>: in flight with the fold
                if (g <= 8) {
                    const int j = tid >> 3, i8 = tid & 7;
                    const double *pj = a.part_md + (size_t)j * HIPK_MAX_PARTS;
                    double hj = 0.0;
                    if (j <= k)   // uniform per 8-lane group
                        hj = hipk_fold_8x8<T>(i8, g, [&](int ci, int ss) { return hipk_peek(pj + ci * kGmSub + ss); });
                    if (i8 == 0) {
                        hj = (j <= k) ? hj : 0.0;
                        hs[j] = hj;
                        rv[j] = ((pass == 0) ? 0.0 : rv[j]) + hj;   // rvec += h (TSL:305), kept by every workgroup
                    }
                } else {   // more than 8 chunks: a wavefront per column, a lane per chunk
                    for (int j = wave; j < HIPK_GM_LDH; j += HIPK_THREADS / 64) {
                        double hj = 0.0;
                        if (j <= k) {   // uniform per wavefront
                            const double *pj = a.part_md + (size_t)j * HIPK_MAX_PARTS;
                            hj = hipk_fold_64x8(lane, g, [&](int ci, int ss) { return hipk_peek(pj + ci * kGmSub + ss); });
                        }
                        if (lane == 0) {
                            hs[j] = hj;
                            rv[j] = ((pass == 0) ? 0.0 : rv[j]) + hj;
                        }
                    }
                }
                if (pass == 0) {
                    double d1 = (double)wt * (double)wt;
                    d1 = hipk_wave_sum(d1);
                    if (lane == 0 && tile < ntiles) hipk_ho_store<LOCAL>(&a.tile_ww[(size_t)tile * 4 + wave], d1);
                }
                HIPK_STAMP(8)
                __syncthreads();
                HIPK_STAMP(9)
                // q = w - V h on the own row; published unnormalised for the gathers of the next SpMV
                {
                    double sacc = 0.0;
                    for (int j0 = 0; j0 <= k; j0 += 8) {   // eight columns' loads in flight, then their links of the chain
                        T vv[8];
                        double hh[8];
#pragma unroll
                        for (int b = 0; b < 8; ++b) {
                            const int jj = (j0 + b <= k) ? j0 + b : k;
                            vv[b] = Vl[(size_t)jj * 256 + tid];
                            hh[b] = hs[jj];
                        }
#pragma unroll
                        for (int b = 0; b < 8; ++b)
                            if (j0 + b <= k) sacc = fma((double)vv[b], hh[b], sacc);
                    }
                    w_own = (T)((double)w_own - sacc);
                    wl[tid] = w_own;
                    if (live) hipk_ho_store<LOCAL>(a.q + row, w_own);
                }
                HIPK_STAMP(10)
                __syncthreads();
                HIPK_STAMP(11)
                if (tid < 32) {   // <q,q>: the chain of virtual thread u, then the 32-lane tree
                    double acc = 0.0;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const double x = (double)wl[e * 32 + tid];
                        acc = fma(x, x, acc);
                    }
                    acc = hipk_half_sum(acc);
                    if (tid == 0) hipk_ho_store<LOCAL>(&a.part_qq[wg], acc);
                }
                HIPK_STAMP(3)
                // second hand-off of the pass; meanwhile thread 192 forms ||rvec|| for the CGS2 decision (TSL:313-326)
                if (hipk_ho_sync<LOCAL>(a.flag_b, wg, nwg, ++seq, 0u, res_lds, [&]() {
                        double rr = 0.0;
                        for (int j0 = 0; j0 <= k; j0 += 8) {
                            double r_[8];
#pragma unroll
                            for (int b = 0; b < 8; ++b) r_[b] = rv[(j0 + b <= k) ? j0 + b : k];
#pragma unroll
                            for (int b = 0; b < 8; ++b)
                                if (j0 + b <= k) rr = fma(r_[b], r_[b], rr);
                        }
                        double rnorm = sqrt(rr < 0.0 ? 0.0 : rr);
                        if (!(rnorm > a.eps)) rnorm = 0.0;
                        bc[3] = rnorm;
                    }) == ~0ull) {
                    if (tid == 0) scal->redo = -1;
                    return;
                }
                HIPK_STAMP(4)
                // ---------------- C: ||q||^2 (every workgroup, same bits); CGS2 decision after the first pass (TSL:313-326)
                if (tid < 64) {
                    const double qq = hipk_fold_64x8(tid, g, [&](int ci, int ss) { return hipk_peek(a.part_qq + ci * kGmSub + ss); });
                    if (tid == 0) {
                        bc[1] = qq;
                        double qnorm = sqrt(qq < 0.0 ? 0.0 : qq);
                        if (!(qnorm > a.eps)) qnorm = 0.0;
                        const double rnorm = bc[3];
                        bc[0] = (rnorm < qnorm * HIPK_INV_SQRT2) ? 1.0 : 0.0;
                    }
                } else if (tid < 128 && pass == 0) {
                    // ||A v||^2 from the tile sums (hipk_fold_tiles8: per chunk the fold of its <= 8 tiles, then the chunks)
                    const double *tp = a.tile_ww;
                    const double ww = hipk_fold_64x8(tid - 64, g, [&](int ci, int tt) {
                        const int tl = ci * (HIPK_BASE_CHUNK / HIPK_TILE) + tt;
                        if (tl >= ntiles) return 0.0;
                        const double *w4 = tp + (size_t)tl * 4;
                        const double w0 = hipk_peek(w4), w1 = hipk_peek(w4 + 1), w2 = hipk_peek(w4 + 2), w3 = hipk_peek(w4 + 3);
                        return 0.0 + ((w0 + w1) + (w2 + w3));
                    });
                    if (tid == 64) bc[2] = ww;
                }
                HIPK_STAMP(12)
                __syncthreads();
                HIPK_STAMP(13)
                if (pass == 1 || bc[0] == 0.0) break;
                __syncthreads();
            }
            // ---------------- normalise: v_{k+1} = q / ||q|| (zero when ||q|| <= eps ||A v_k||), TSL:358-387
            const double qq = bc[1], ww = bc[2];
            double norm1 = sqrt(qq < 0.0 ? 0.0 : qq);
            double norm0 = sqrt(ww < 0.0 ? 0.0 : ww);
            if (!(norm0 > a.eps)) norm0 = 0.0;
            const double thr = a.eps * norm0;
            const bool use = norm1 > thr;
            const T nrm = (T)norm1;
            const T vnew = (use && live) ? w_own / nrm : (T)0;
            if (k + 1 < m) Vl[(size_t)(k + 1) * 256 + tid] = vnew;
            if (live) wcol[row] = vnew;              // only read by launches after this one (the host's fallback path)
            from_q = true;
            nrm_prev = nrm;
            use_prev = use;
            // column k of H, breakdown, 'incremental': Givens update + early exit (TSL:358-387, 595-623): EVERY workgroup, from
            // its own copies -- the same bits everywhere, so the loop test needs no exchange; workgroup 0 also writes memory
            if (!use) norm1 = 0.0;
            if (!a.incremental) {
                if (tid <= k) HRl[tid * HIPK_GM_LDH + k] = rv[tid];
                if (tid == 0) HRl[(k + 1) * HIPK_GM_LDH + k] = norm1;
            }
            if (wg == 0) {
                if (tid <= k) scal->H[tid * HIPK_GM_LDH + k] = rv[tid];
                if (tid == 0) {
                    scal->H[(k + 1) * HIPK_GM_LDH + k] = norm1;
                    scal->steps_done = k + 1;
                }
            }
            if (tid == 0) {
                bool stp = false;
                if (norm1 == 0.0) {  // TSL:387
                    if (wg == 0) scal->breakdown = 1;
                    bc[5] = 1.0;
                    stp = true;
                }
                if (a.incremental) {
                    // the rotations of the earlier steps live in LDS (gvl); eight at a time: loads first, then the dependent chain
                    double cur = rv[0];   // hc[i] under rotation i; hc[i+1] is still rvec[i+1] when rotation i reads it
                    for (int i0 = 0; i0 < k; i0 += 8) {
                        double cs8[8], sn8[8], nx8[8];
#pragma unroll
                        for (int b = 0; b < 8; ++b) {
                            const int ii = (i0 + b < k) ? i0 + b : k - 1;
                            cs8[b] = gvl[2 * ii];
                            sn8[b] = gvl[2 * ii + 1];
                            nx8[b] = rv[ii + 1];
                        }
#pragma unroll
                        for (int b = 0; b < 8; ++b)
                            if (i0 + b < k) {
                                const double cs = cs8[b], sn = sn8[b];
                                const double p0 = cs * cur, p1 = sn * nx8[b];
                                const double t0 = p0 - p1;
                                const double p2 = sn * cur, p3 = cs * nx8[b];
                                HRl[(i0 + b) * HIPK_GM_LDH + k] = t0;
                                if (wg == 0) scal->R[(i0 + b) * HIPK_GM_LDH + k] = t0;
                                cur = p2 + p3;
                            }
                    }
                    double hk = cur;
                    const double hk1 = norm1;
                    double cs, sn;
                    hipk_givens(hk, hk1, cs, sn);
                    gvl[2 * k] = cs;
                    gvl[2 * k + 1] = sn;
                    {
                        const double p0 = cs * hk, p1 = sn * hk1;
                        hk = p0 - p1;
                    }
                    HRl[k * HIPK_GM_LDH + k] = hk;
                    const double b0 = bvl[k], b1 = bvl[k + 1];
                    const double p0 = cs * b0, p1 = sn * b1;
                    const double t0 = p0 - p1;
                    const double p2 = sn * b0, p3 = cs * b1;
                    const double bk1 = p2 + p3;
                    bvl[k] = t0;
                    bvl[k + 1] = bk1;
                    bvl[k + 2] = 0.0;
                    const double err = fabs(bk1);
                    if (wg == 0) {
                        scal->gv[2 * k] = cs;
                        scal->gv[2 * k + 1] = sn;
                        scal->R[k * HIPK_GM_LDH + k] = hk;
                        scal->beta_vec[k] = t0;
                        scal->beta_vec[k + 1] = bk1;
                        scal->err = err;
                    }
                    if (!(err > a.ptol)) stp = true;  // TSL:591
                }
                bc[6] = stp ? 1.0 : 0.0;
            }
            HIPK_STAMP(14)
            __syncthreads();                         // Vl[k+1], HRl, bc[6]
            HIPK_STAMP(5)
            if (bc[6] != 0.0) {
                kk = k + 1;
                break;
            }
        }
        // ================ end of the cycle (TSL:754-764) ================
        HIPK_STAMP(5)
        matvecs += kk;
        if (bc[5] != 0.0) breakdown_any = 1;
        bool solved;
        if (a.incremental) {
            hipk_gm_trisolve_lds(sm, kk, tid);
            solved = true;
        } else {
            solved = hipk_gm_lstsq_lds(sm, kk, beta0, tid);
        }
        HIPK_STAMP(6)
        if (!solved) {   // uniform (every workgroup factorises the same bits): the host finishes this cycle
            matvecs -= kk;
            status = 2;
            break;
        }
        // x += V[:, :kk] y on the own row (TSL:488-490), published for the residual's gathers
        {
            double sacc = 0.0;
            for (int j0 = 0; j0 < kk; j0 += 8) {
                T vv[8];
                double yy[8];
#pragma unroll
                for (int b = 0; b < 8; ++b) {
                    const int jj = (j0 + b < kk) ? j0 + b : kk - 1;
                    vv[b] = Vl[(size_t)jj * 256 + tid];
                    yy[b] = yl[jj];
                }
#pragma unroll
                for (int b = 0; b < 8; ++b)
                    if (j0 + b < kk) sacc = fma((double)vv[b], yy[b], sacc);
            }
            x_own = (T)((double)x_own + sacc);
            if (live) hipk_ho_store<LOCAL>(a.x + row, x_own);
        }
        HIPK_HO(a.flag_a)
        // r = (M)(b - A x) on the own row (TSL:791), unnormalised into a.q
        T r_own;
        {
            T xs[kGmRowRegs];
#pragma unroll
            for (int j = 0; j < kGmRowRegs; ++j)
                if (j < wmax) xs[j] = hipk_peek_off<T>(a.x, cj[j]);
            T acc_row = (T)0;
#pragma unroll
            for (int j = 0; j < kGmRowRegs; ++j)
                if (j < wmax) {
                    const T p = vj[j] * xs[j];
                    acc_row = (j < len) ? acc_row + p : acc_row;
                }
            for (int j = kGmRowRegs; j < len; ++j) {
                const T p = a.val[lo + j] * hipk_peek_t<T>(a.x + a.col[lo + j]);
                acc_row = acc_row + p;
            }
            r_own = b_own - acc_row;
            if (a.dscale) r_own = dsc * r_own;
            if (!live) r_own = (T)0;
            if (live) hipk_ho_store<LOCAL>(a.q + row, r_own);
        }
        matvecs += 1;
        HIPK_HO(a.flag_b)
        {   // wavefront sums of <r,r> over the own tile
            T rt = (T)0;
            if (trow < n) rt = hipk_peek_t<T>(a.q + trow);
            double d1 = (double)rt * (double)rt;
            d1 = hipk_wave_sum(d1);
            if (lane == 0 && tile < ntiles) hipk_ho_store<LOCAL>(&a.tile_ww[(size_t)tile * 4 + wave], d1);
        }
        HIPK_HO(a.flag_a)
        if (tid < 64) {
            const double *tp = a.tile_ww;
            const double r2 = hipk_fold_64x8(tid, g, [&](int ci, int tt) {
                const int tl = ci * (HIPK_BASE_CHUNK / HIPK_TILE) + tt;
                if (tl >= ntiles) return 0.0;
                const double *w4 = tp + (size_t)tl * 4;
                const double w0 = hipk_peek(w4), w1 = hipk_peek(w4 + 1), w2 = hipk_peek(w4 + 2), w3 = hipk_peek(w4 + 3);
                return 0.0 + ((w0 + w1) + (w2 + w3));
            });
            if (tid == 0) bc[7] = r2;
        }
        __syncthreads();
        {   // unit residual + norm (`_safe_normalize`, TSL:217-273)
            const double res2 = bc[7];
            const double norm = sqrt(res2 < 0.0 ? 0.0 : res2);
            const bool use = norm > a.eps;
            const T nrm = (T)norm;
            const T v0 = (use && live) ? r_own / nrm : (T)0;
            Vl[tid] = v0;
            if (live) a.V[row] = v0;                 // only read by launches after this one
            from_q = true;
            nrm_prev = nrm;
            use_prev = use;
            beta0 = use ? norm : 0.0;
        }
        ++cycles;
        __syncthreads();
        HIPK_STAMP(7)
        if (!(cycles < a.cycles_left && beta0 > a.atol_eff)) {
            status = 1;
            break;
        }
        if (cycles >= a.max_cycles) break;           // status 0: the host launches again
    }
    if (wg == 0 && tid == 0) {
        scal->res_norm = beta0;
        scal->rep_cycles = cycles;
        scal->rep_matvecs = matvecs;
        scal->rep_status = status;
        scal->rep_breakdown = breakdown_any;
    }
#ifdef HIPK_GM_STAMPS
    if (stamping)
        for (int i = 0; i < 16; ++i) a.stamps[i] += t_acc[i];
#endif
#undef HIPK_STAMP
#undef HIPK_HO
}

// after a speculation miss at step k (hipk_gm_normalize_kernel, guard): steps >= k of the cycle are enqueued again
__global__ void hipk_gm_resume_kernel(hipk_gm_scal *__restrict__ scal) {
    scal->stop_step = INT64_MAX;
    scal->redo = 0;
    scal->redo_step = -1;
    scal->pass2 = 0;
}

#include "hipk_gm_mid.h"   // the Arnoldi steps of a cycle in one launch, mid-size systems (uses hipk_gm_scal, hipk_gm_want_pass2, hipk_givens)

// big: the workspace block of a solve with restart m > 31 (hipk_gm_big_doubles(m) doubles), else null: the struct's own arrays
__global__ void hipk_gm_cycle_init_kernel(hipk_gm_scal *__restrict__ scal, int incremental, double ptol, double *big = nullptr,
                                          int m = HIPK_GM_MAXM) {
    const int t = threadIdx.x;
    hipk_gm_view v;
    if (big != nullptr) {
        const int ld = ((m + 1 + 7) / 8) * 8;
        v.ldh = ld;
        v.m = m;
        v.H = big;
        v.R = v.H + (size_t)(m + 2) * ld;
        v.gv = v.R + (size_t)ld * ld;
        v.beta_vec = v.gv + 2 * ld;
        v.hvec = v.beta_vec + (ld + 1);
        v.rvec = v.hvec + ld;
    } else {
        v.ldh = HIPK_GM_LDH;
        v.m = HIPK_GM_MAXM;
        v.H = scal->H;
        v.R = scal->R;
        v.gv = scal->gv;
        v.beta_vec = scal->beta_vec;
        v.hvec = scal->hvec;
        v.rvec = scal->rvec;
    }
    const int ld = v.ldh;
    for (int i = t; i < (v.m + 2) * ld; i += blockDim.x) v.H[i] = 0.0;
    for (int i = t; i < ld * ld; i += blockDim.x) v.R[i] = ((i / ld) == (i % ld)) ? 1.0 : 0.0;  // TSL:581
    for (int i = t; i < ld * 2; i += blockDim.x) v.gv[i] = 0.0;
    for (int i = t; i <= ld; i += blockDim.x) v.beta_vec[i] = (i == 0) ? scal->res_norm : 0.0;
    if (t == 0) {
        scal->v = v;
        scal->bar = 0;
        scal->xcc_mask = 0;
        scal->redo = 0;
        scal->redo_step = -1;
        scal->stop_step = INT64_MAX;
        scal->steps_done = 0;
        scal->pass2 = 0;
        scal->breakdown = 0;
        scal->incremental = incremental;
        scal->ptol = ptol;
        scal->err = scal->res_norm;
    }
}

// x += V[:, :k] y   (TSL:488-490)
template <typename T, int KC = HIPK_GM_LDH>
__global__ __launch_bounds__(HIPK_THREADS) void hipk_gm_xupdate_kernel(int64_t n, int ch, int k,
                                                                       const T *__restrict__ V, int64_t ldv,
                                                                       T *__restrict__ x, hipk_gm_yN<KC> yy) {
    hipk_chunk_loop<T>(n, ch, blockIdx.x, [&](int64_t i, int nv) {
        constexpr int VEC = hipk_vec<T>::VEC;
        T xv[VEC];
        hipk_ld<T>((const T *)x, i, nv, xv);
        double s[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) s[e] = 0.0;
        if constexpr (KC <= HIPK_GM_LDH) {
#pragma unroll
            for (int j = 0; j < KC; ++j) {
                if (j < k) {
                    T vv[VEC];
                    hipk_ld_nt_vec<T>(V + (int64_t)j * ldv, i, nv, vv);   // every column is read once: stream it
#pragma unroll
                    for (int e = 0; e < VEC; ++e) s[e] = fma((double)vv[e], yy.y[j], s[e]);
                }
            }
        } else {   // restart > 31: the same chain, four column loads in flight
            int j = 0;
            for (; j + 4 <= k; j += 4) {
                T vv[4][VEC];
#pragma unroll
                for (int b = 0; b < 4; ++b) hipk_ld_nt_vec<T>(V + (int64_t)(j + b) * ldv, i, nv, vv[b]);
#pragma unroll
                for (int b = 0; b < 4; ++b)
#pragma unroll
                    for (int e = 0; e < VEC; ++e) s[e] = fma((double)vv[b][e], yy.y[j + b], s[e]);
            }
            for (; j < k; ++j) {
                T vv[VEC];
                hipk_ld_nt_vec<T>(V + (int64_t)j * ldv, i, nv, vv);
#pragma unroll
                for (int e = 0; e < VEC; ++e) s[e] = fma((double)vv[e], yy.y[j], s[e]);
            }
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) xv[e] = (T)((double)xv[e] + s[e]);
        hipk_st<T>(x, i, nv, xv);
    });
}

// residual (already in column 0) -> unit residual + norm  (`_safe_normalize`, TSL:217-273); also <b,b>
template <typename T>
__global__ __launch_bounds__(HIPK_THREADS) void hipk_gm_resnorm_kernel(int64_t n, int ch, int g,
                                                                       hipk_gm_scal *__restrict__ scal,
                                                                       T *__restrict__ v0,
                                                                       const double *__restrict__ part_res,
                                                                       const double *__restrict__ part_bb, double eps) {
    __shared__ double sbuf[2 * HIPK_THREADS];
    double res2, bs;
    hipk_reduce_parts2(part_res, part_bb, g, res2, bs, sbuf);
    const double norm = sqrt(res2 < 0.0 ? 0.0 : res2);
    const bool use = norm > eps;
    const T nrm = (T)norm;
    hipk_chunk_loop<T>(n, ch, blockIdx.x, [&](int64_t i, int nv) {
        constexpr int VEC = hipk_vec<T>::VEC;
        T rv[VEC];
        hipk_ld<T>((const T *)v0, i, nv, rv);
#pragma unroll
        for (int e = 0; e < VEC; ++e) rv[e] = use ? rv[e] / nrm : (T)0;
        hipk_st<T>(v0, i, nv, rv);
    });
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        scal->res_norm = use ? norm : 0.0;
        scal->bs = bs;
    }
}

__global__ __launch_bounds__(HIPK_THREADS) void hipk_gm_final_kernel(hipk_gm_scal *__restrict__ scal, int g,
                                                                     const double *__restrict__ part_res,
                                                                     const double *__restrict__ part_xx) {
    __shared__ double sbuf[2 * HIPK_THREADS];
    double res2, xx;
    hipk_reduce_parts2(part_res, part_xx, g, res2, xx, sbuf);
    if (threadIdx.x == 0) {
        scal->res2 = res2;
        scal->xx = xx;
    }
}

// ---------------------------------------------------------------- host-side small dense solves
// `_lstsq` (TSL:391-428): normal equations + Cholesky, general solve when Cholesky fails.
// Same operation order as oracle/krylov_oracle.c::lstsq_normal.
static void hipk_lstsq_normal(const double *H, int ldh, int k, double beta0, double *y) {
    // leading dimension LD of the scratch arrays: 32 up to restart 31 (as ever), k beyond; the arithmetic does not depend on it
    const int LD = k <= 32 ? 32 : k;
    std::vector<double> a2v((size_t)LD * LD), b2v(LD), Lv((size_t)LD * LD, 0.0), zv(LD), Mv((size_t)LD * (LD + 1));
    double *a2 = a2v.data(), *b2 = b2v.data(), *L = Lv.data(), *z = zv.data(), *M = Mv.data();
    for (int i = 0; i < k; ++i) {
        for (int j = 0; j < k; ++j) {
            double s = 0.0;
            for (int p = 0; p <= k; ++p) s = fma(H[p * ldh + i], H[p * ldh + j], s);
            a2[i * LD + j] = s;
        }
        b2[i] = H[0 * ldh + i] * beta0;
    }
    bool ok = true;
    for (int j = 0; j < k && ok; ++j) {
        double d = a2[j * LD + j];
        for (int p = 0; p < j; ++p) d = fma(-L[j * LD + p], L[j * LD + p], d);
        if (!(d > 0.0)) {
            ok = false;
            break;
        }
        const double ljj = sqrt(d);
        L[j * LD + j] = ljj;
        for (int i = j + 1; i < k; ++i) {
            double s = a2[i * LD + j];
            for (int p = 0; p < j; ++p) s = fma(-L[i * LD + p], L[j * LD + p], s);
            L[i * LD + j] = s / ljj;
        }
    }
    if (ok) {
        for (int i = 0; i < k; ++i) {
            double s = b2[i];
            for (int p = 0; p < i; ++p) s = fma(-L[i * LD + p], z[p], s);
            z[i] = s / L[i * LD + i];
        }
        for (int i = k - 1; i >= 0; --i) {
            double s = z[i];
            for (int p = i + 1; p < k; ++p) s = fma(-L[p * LD + i], y[p], s);
            y[i] = s / L[i * LD + i];
        }
        return;
    }
    for (int i = 0; i < k; ++i) {
        for (int j = 0; j < k; ++j) M[i * (LD + 1) + j] = a2[i * LD + j];
        M[i * (LD + 1) + k] = b2[i];
    }
    for (int c = 0; c < k; ++c) {
        int piv = c;
        for (int i = c + 1; i < k; ++i)
            if (fabs(M[i * (LD + 1) + c]) > fabs(M[piv * (LD + 1) + c])) piv = i;
        if (piv != c)
            for (int j = 0; j <= k; ++j) {
                const double tmp = M[c * (LD + 1) + j];
                M[c * (LD + 1) + j] = M[piv * (LD + 1) + j];
                M[piv * (LD + 1) + j] = tmp;
            }
        for (int i = c + 1; i < k; ++i) {
            const double f = M[i * (LD + 1) + c] / M[c * (LD + 1) + c];
            for (int j = c; j <= k; ++j) M[i * (LD + 1) + j] = fma(-f, M[c * (LD + 1) + j], M[i * (LD + 1) + j]);
        }
    }
    for (int i = k - 1; i >= 0; --i) {
        double s = M[i * (LD + 1) + k];
        for (int p = i + 1; p < k; ++p) s = fma(-M[i * (LD + 1) + p], y[p], s);
        y[i] = s / M[i * (LD + 1) + i];
    }
}

static inline double hipk_tmin(double a, double b) {
    if (isnan(a) || isnan(b)) return NAN;
    return a < b ? a : b;
}

extern "C" size_t hipk_gmres_work_bytes(int64_t n, int restart, int dtype) {
    const size_t sv = (dtype == HIPK_F64) ? 8 : 4;
    const size_t vec = hipk_align_up((size_t)(n > 0 ? n : 1) * sv, 256);
    const int m = restart < 1 ? 1 : (restart > HIPK_GM_MAXM_BIG ? HIPK_GM_MAXM_BIG : restart);
    // mid-size systems (hipk_gm_mid.h): v_{k+1} as 16-byte flagged words (two more vectors) + the partial slots
    const hipk_geom gm = hipk_make_geom(n > 0 ? n : 1);
    const bool mid = gm.g > 8 && gm.g <= kGmMidMaxChunks && m <= HIPK_GM_MAXM;
    return kGmHeader + hipk_gm_big_doubles(m) * sizeof(double) + (size_t)(kGmSlots + m + 1) * HIPK_MAX_PARTS * sizeof(double) +
           (size_t)(m + 2) * vec + (mid ? hipk_align_up((size_t)(n > 0 ? n : 1) * 16, 256) + kGmMidSlotBytes : 0);   // v_{k+1} as 16-byte flagged words
}

// partials of || d .* v ||^2 (||M b|| of the preconditioned solver, TSL:750)
template <typename T>
__global__ __launch_bounds__(HIPK_THREADS) void hipk_gm_scaled_sq_kernel(int64_t n, int ch, const T *__restrict__ v,
                                                                         const T *__restrict__ d,
                                                                         double *__restrict__ part) {
    __shared__ double sbuf[HIPK_THREADS];
    const int c = blockIdx.x;
    double acc = 0.0;
    hipk_chunk_loop<T>(n, ch, c, [&](int64_t i, int nv) {
        constexpr int VEC = hipk_vec<T>::VEC;
        T vv[VEC], dv[VEC];
        hipk_ld<T>(v, i, nv, vv);
        hipk_ld<T>(d, i, nv, dv);
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            const T m = dv[k] * vv[k];
            if (k < nv) acc = fma((double)m, (double)m, acc);
        }
    });
    acc = hipk_block_sum(acc, sbuf);
    if (threadIdx.x == 0) part[c] = acc;
}

// dinv != nullptr: left Jacobi preconditioning -- every A(.) is followed by M(.) = dinv .* (.) (TSL:351, 791, 766), applied
// by the SpMV epilogue (HIPK_SPMV_SCALE) before its fused dots; ptol from ||M b|| (TSL:750).  Mirrored by orc_gmres_jacobi.
// cb != nullptr (dinv == nullptr): M is the CALLER's device code, cb(user, in, out) enqueues out = M(in) on `stream`.  It runs
// in place on the vector the SpMV just wrote; the squared norm the Jacobi form fuses into the SpMV epilogue is then a
// chunked dot of the preconditioned vector (one more pass over it).
template <typename T>
static int hipk_gmres_solve_t(hipk_csr_s *A, const T *dinv, const T *b, T *x, char *work, const hipk_params *prm,
                              hipk_stats *st, hipStream_t stream, hipk_precond_fn cb = nullptr, void *user = nullptr) {
    const bool ext = cb != nullptr;
    const int64_t n = A->n_rows;
    const hipk_geom gm = A->geom;
    const int m = prm->restart;
    const size_t vec = hipk_align_up((size_t)n * sizeof(T), 256);
    const int64_t ldv = (int64_t)(vec / sizeof(T));
    hipk_gm_scal *scal = (hipk_gm_scal *)work;
    // restart > 31: H, R, the Givens pairs, beta, h and r live in a block of the workspace behind the header (hipk_gm_view)
    const size_t big_n = hipk_gm_big_doubles(m);
    double *big = big_n ? (double *)(work + kGmHeader) : nullptr;
    double *parts = (double *)(work + kGmHeader + big_n * sizeof(double));
    double *part_ww = parts, *part_qq = parts + HIPK_MAX_PARTS, *part_res = parts + 2 * HIPK_MAX_PARTS;
    double *part_bb = parts + 3 * HIPK_MAX_PARTS, *part_xx = parts + 4 * HIPK_MAX_PARTS;
    double *part_spare = parts + 5 * HIPK_MAX_PARTS;
    double *part_md = parts + (size_t)kGmSlots * HIPK_MAX_PARTS;
    char *vbase = (char *)parts + (size_t)(kGmSlots + m + 1) * HIPK_MAX_PARTS * sizeof(double);
    T *V = (T *)vbase;
    T *tmp = (T *)(vbase + (size_t)(m + 1) * vec);
    const int incremental = (prm->gmres_method == HIPK_GMRES_INCREMENTAL) ? 1 : 0;
    // guards (`_safe_normalize`, breakdown threshold) use the eps of the working dtype, as torch.finfo(dtype) would
    const double eps_t = (sizeof(T) == 8) ? HIPK_EPS64 : HIPK_EPS32;
    const int64_t maxiter = (prm->maxiter < 0) ? 10 * n : prm->maxiter;  // TSL:719-721

    hipk_event_pair whole;
    HIPK_CHECK_HIP(whole.create());
    hipk_spmv_profiler prof(prm->profile != 0 ? HIPK_K_SPMV : 0);
    HIPK_CHECK_HIP(hipEventRecord(whole.a, stream));

    hipk_spmv_args sa;
    memset(&sa, 0, sizeof(sa));
    sa.crow = A->crow;
    sa.col = A->col;
    sa.val = A->val;
    sa.n = n;
    sa.ch = gm.ch;
    sa.g = gm.g;
    sa.dscale = dinv;
    const int scale_bit = dinv ? HIPK_SPMV_SCALE : 0;
    int rc;
    int64_t matvecs = 0;
    // ext: v <- M(v) by the caller, part[c] = chunk partials of <v,v>
    auto precondition = [&](T *v, double *part) -> int {
        if (cb(user, v, v) != 0) {
            hipk_set_error("hipk_pgmres_solve_cb: the preconditioner callback failed");
            return HIPK_ERR_ARG;
        }
        return hipk_launch_dot_parts(n, v, v, A->dtype, part, stream);
    };

    // residual = M(b - A x0) into column 0, unit residual + norm (TSL:791-792); <b,b>
    hipk_spmv_args sr = sa;
    sr.x = x;
    sr.y = V;
    sr.mode = HIPK_SPMV_RESID | HIPK_SPMV_DOT_YY | scale_bit;
    sr.bsub = b;
    sr.part0 = part_spare;
    sr.part1 = part_res;
    if ((rc = hipk_launch_dot_parts(n, b, b, A->dtype, part_bb, stream)) != HIPK_OK) return rc;
    if ((rc = hipk_launch_spmv(A, sr, stream)) != HIPK_OK) return rc;
    ++matvecs;
    if (ext && (rc = precondition(V, part_res)) != HIPK_OK) return rc;
    hipk_gm_resnorm_kernel<T><<<gm.g, HIPK_THREADS, 0, stream>>>(n, gm.ch, gm.g, scal, V, part_res, part_bb, eps_t);
    HIPK_CHECK_HIP(hipGetLastError());
    double head[2];
    HIPK_CHECK_HIP(hipMemcpyAsync(head, scal, sizeof(head), hipMemcpyDeviceToHost, stream));
    HIPK_CHECK_HIP(hipStreamSynchronize(stream));
    double res_norm = head[0];
    const double bs = head[1];
    const double b_norm = hipk_norm_from_sq(bs);

    // TSL:735-753 (python floats become fp32 tensors; python max() keeps a float a float)
    const double eps = HIPK_EPS64;  // the absolute floor keeps the fp64 eps also for fp32 storage
    const double sq = sqrt((double)n);
    const double cand = (prm->gpu_tolerances ? 1e-12 : 1e-14) * sq;
    const double adaptive = (cand > prm->tol) ? cand : (double)(float)prm->tol;
    const double base_atol = (double)(float)(eps * (prm->gpu_tolerances ? 1000 : 100) * (double)n);
    const double atol_eff = hipk_tmax(adaptive * b_norm, hipk_tmax((double)(float)prm->atol, base_atol));
    double mb_norm = b_norm;  // ||M b|| (TSL:750)
    if (dinv || ext) {
        if (ext) {  // M b through the spare vector (the callback sees workspace vectors only)
            HIPK_CHECK_HIP(hipMemcpyAsync(tmp, b, (size_t)n * sizeof(T), hipMemcpyDeviceToDevice, stream));
            if ((rc = precondition(tmp, part_spare)) != HIPK_OK) return rc;
        } else {
            hipk_gm_scaled_sq_kernel<T><<<gm.g, HIPK_THREADS, 0, stream>>>(n, gm.ch, b, dinv, part_spare);
        }
        double *mb_dev = part_xx;  // the <x,x> slot is unused until the end of the solve
        if ((rc = hipk_launch_finish1(part_spare, gm.g, mb_dev, stream)) != HIPK_OK) return rc;
        double mb2 = 0.0;
        HIPK_CHECK_HIP(hipMemcpyAsync(&mb2, mb_dev, sizeof(double), hipMemcpyDeviceToHost, stream));
        HIPK_CHECK_HIP(hipStreamSynchronize(stream));
        mb_norm = hipk_norm_from_sq(mb2);
    }
    const double ptol = mb_norm * hipk_tmin(1.0, atol_eff / b_norm);

    hipk_gm_scal *hs = (hipk_gm_scal *)malloc(sizeof(hipk_gm_scal));
    if (!hs) {
        hipk_set_error("out of host memory");
        return HIPK_ERR_ARG;
    }
    std::vector<double> big_host(big_n);
    int64_t cycles = 0;
    // launch-bound systems: fewer launches per step (their kernels keep H in the struct's 32-wide arrays: restart <= 31)
    const bool small = gm.g <= 8 && m <= HIPK_GM_MAXM && !getenv("HIPK_GMRES_NO_SMALL");
    const bool wide = small && gm.ch == HIPK_BASE_CHUNK && !getenv("HIPK_GMRES_NO_WIDE");  // hipk_gm_update_wide_kernel
    // small systems with short rows: the whole restart cycle in ONE launch (hipk_gm_cycle_small_kernel)
    bool cyc = wide && !ext && A->max_row_len <= HIPK_LONG_ROW && !getenv("HIPK_GMRES_NO_CYCLE");
    // ... with the basis in LDS and eight workgroups per chunk (hipk_gm_solve_lds_kernel) when m columns of 256 rows fit
    // two workgroups share a compute unit's 160 KB of LDS
    // ... and 8 g workgroups fit ONE XCD (an eighth of the compute units, two workgroups each); a process whose resident
    // workgroups once failed to meet (a shared device) does not try again: the wait for that verdict takes seconds
    static bool lds_cycle_failed = false;
    // 9 .. 32 chunks (n <= 65536): the same kernel with its workgroups spread over the chip (agent-scope hand-offs)
    const bool lds_spread = gm.g > 8 && gm.g <= 32 && m <= HIPK_GM_MAXM && gm.ch == HIPK_BASE_CHUNK && !ext && A->max_row_len <= HIPK_LONG_ROW &&
                            !getenv("HIPK_GMRES_NO_SMALL") && !getenv("HIPK_GMRES_NO_CYCLE") && !getenv("HIPK_NO_LDS_SPREAD");
    bool cyc_lds = (cyc || lds_spread) && hipk_gm_solve_lds_bytes<T>(m) <= 80 * 1024 &&
                   kGmSub * gm.g <= (lds_spread ? 2 * A->n_cu : 2 * (A->n_cu / 8)) && !lds_cycle_failed && !getenv("HIPK_GMRES_NO_LDS_CYCLE");
    if (lds_spread) cyc = cyc_lds;   // beyond 8 chunks there is no one-workgroup-per-chunk kernel to fall back to
    if (cyc_lds) {
        static bool attr_done[2] = {false, false};   // the launches ask for more than the default 64 KB of dynamic LDS
        if (!attr_done[sizeof(T) == 8]) {
            (void)hipFuncSetAttribute((const void *)hipk_gm_solve_lds_kernel<T, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
            (void)hipFuncSetAttribute((const void *)hipk_gm_solve_lds_kernel<T, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
            attr_done[sizeof(T) == 8] = true;
        }
    }
    // its hand-offs through the shared L2 of ONE XCD (plain stores; placement verified by the kernel), else agent-scope stores
    bool cyc_local = !lds_spread && !getenv("HIPK_GM_CYCLE_AGENT");
    auto env_int = [](const char *name, int dflt) { const char *e = getenv(name); return e ? atoi(e) : dflt; };
    const bool stream_k = !small && (m > HIPK_GM_MAXM || !getenv("HIPK_GMRES_NO_STREAM"));  // large systems: hipk_gm_*_stream_kernel
    // 33 .. 256 chunks (fp64, no preconditioner, restart <= 31, rows of <= 12 entries within a window around their chunk): the
    // Arnoldi steps of a cycle in ONE launch, one workgroup per chunk (hipk_gm_mid.h); HIPK_GMRES_MID=0 keeps the launches
    static bool mid_failed = false;
    bool mid_cycle = false;
    hipk_mid_plan mid_plan;
    memset(&mid_plan, 0, sizeof(mid_plan));
    size_t mid_lds = 0;
    void (*mid_kern)(hipk_gm_mid_args) = nullptr;
    {
        const int mid_min = env_int("HIPK_GMRES_MID_MIN", kGmMidMinChunks);   // (A/B against the whole-solve kernel of 9 .. 32 chunks)
        mid_cycle = !small && !ext && m <= HIPK_GM_MAXM && gm.g > (mid_min < 8 ? 8 : mid_min) && gm.g <= kGmMidMaxChunks &&
                    gm.g <= A->n_cu && gm.ch == HIPK_BASE_CHUNK && A->op_cb == nullptr && A->crow != nullptr && A->max_row_len <= 12 &&
                    prm->profile == 0 && !mid_failed && !(getenv("HIPK_GMRES_MID") && getenv("HIPK_GMRES_MID")[0] == '0') &&
                    !getenv("HIPK_GMRES_NO_CYCLE");
        if (dinv)
            mid_kern = A->max_row_len <= 5 ? hipk_gm_mid_kernel<T, 5, true> : A->max_row_len <= 7 ? hipk_gm_mid_kernel<T, 7, true>
                       : A->max_row_len <= 9 ? hipk_gm_mid_kernel<T, 9, true> : hipk_gm_mid_kernel<T, 12, true>;
        else
            mid_kern = A->max_row_len <= 5 ? hipk_gm_mid_kernel<T, 5> : A->max_row_len <= 7 ? hipk_gm_mid_kernel<T, 7>
                       : A->max_row_len <= 9 ? hipk_gm_mid_kernel<T, 9> : hipk_gm_mid_kernel<T, 12>;
        if (mid_cycle) {
            mid_cycle = hipk_mid_plan_get(A, 1, stream, &mid_plan);   // the tiles each workgroup's window holds (hipk_mid.h)
            mid_lds = mid_cycle ? hipk_gm_mid_lds_bytes(mid_plan.max_slots * HIPK_TILE, sizeof(T)) : 0;
            int occ = 0;
            mid_cycle = mid_cycle && mid_plan.max_slots <= kMidPlanSlots && mid_lds <= (size_t)160 * 1024 &&
                        hipFuncSetAttribute((const void *)mid_kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mid_lds) == hipSuccess &&
                        hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, mid_kern, 1024, mid_lds) == hipSuccess && (int64_t)occ * A->n_cu >= gm.g;
            (void)hipGetLastError();
        }
        if (mid_cycle) cyc = cyc_lds = false;   // (9 .. 32 chunks: instead of the whole-solve kernel)
    }
    // multi-dot with up to 32 columns per workgroup (w read ONCE per step; 0, the default: groups of 8, w re-read per group).  Same
    // box, alternating, N = 4 M (profiles/r03_gmres_history.md): GMRES(30) 6.67-6.70 vs 6.64-6.66 ms per cycle, GMRES(50) 16.50 vs
    // 16.38, GMRES(100) 60.1-60.2 vs 59.5-59.8 -- the re-reads of w are Infinity-Cache hits (FETCH_SIZE counts them: the 1.15 x of
    // VERDICT r2), and 113 VGPRs (4 workgroups per CU) cost more than they did.  Kept for the counters, not taken.
    const bool md_wide = env_int("HIPK_GM_MD_WIDE", 0) != 0;
    // normalise step as hipk_gm_hcol_kernel + hipk_gm_scale_kernel (HIPK_GM_SPLIT_NORM=0: the one-kernel form)
    // from kGmSplitChunks reduction chunks up: below, a workgroup's fold of 2 g partials is cheap and the extra launch is not
    const bool split_norm = !small && env_int("HIPK_GM_SPLIT_NORM", gm.g >= kGmSplitChunks ? 1 : 0) != 0;
    const int gm_nres = env_int("HIPK_GM_NRES", 5);  // basis columns read with the default cache policy (the rest: nt)
    // large systems: second-pass launches only at the steps where a second CGS pass is expected (step 0, then every step
    // that ever asked for one in this solve); a miss is caught on the device and the cycle re-enqueued from that step
    const bool spec = !small && env_int("HIPK_GM_SPEC", 1) != 0;
    bool predict[HIPK_GM_MAXM_BIG + 1];
    for (int j = 0; j <= HIPK_GM_MAXM_BIG; ++j) predict[j] = (j == 0) && env_int("HIPK_GM_SPEC", 1) != 2;  // 2: learn everything (tests)
    int happy = 0;
    int lds_launch_no = 0;
    int64_t prof_valid = 0;
    rc = HIPK_OK;
    const int nt = (int)((n + HIPK_TILE - 1) / HIPK_TILE);
    while (cycles < maxiter && res_norm > atol_eff) {
        hipk_gm_cycle_init_kernel<<<1, HIPK_THREADS, 0, stream>>>(scal, incremental, ptol, big, m);
        int k_start = 0;
      enqueue:
        if (cyc) {
            hipk_gm_cyc_args<T> ca;
            ca.n = n;
            ca.g = gm.g;
            ca.m = m;
            ca.scal = scal;
            ca.V = V;
            ca.ldv = ldv;
            ca.crow = A->crow;
            ca.col = A->col;
            ca.val = (const T *)A->val;
            ca.dscale = dinv;
            ca.part_md = part_md;
            ca.part_qq = part_qq;
            ca.tile_ww = A->tile_part + 4 * (size_t)nt;
            ca.q = tmp;
            ca.incremental = incremental;
            ca.ptol = ptol;
            ca.beta0 = res_norm;
            ca.b = b;
            ca.x = x;
            ca.atol_eff = atol_eff;
            ca.cycles_left = maxiter - cycles;
            ca.max_cycles = env_int("HIPK_GM_LAUNCH_CYCLES", 64);
            {   // tests: HIPK_TEST_LDS_NOT_RESIDENT=k makes the k-th one-launch kernel of this solve report "not co-resident"
                const char *fe = getenv("HIPK_TEST_LDS_NOT_RESIDENT");
                const int fail_launch = fe ? (atoi(fe) > 1 ? atoi(fe) : 1) : 0;
                ca.test_not_resident = (++lds_launch_no == fail_launch) ? 1 : 0;
            }
            ca.bar = &scal->bar;
            ca.eps = eps_t;
            ca.stamps = getenv("HIPK_GM_STAMPS") ? (unsigned long long *)(part_spare + 1600) : nullptr;
            if (ca.stamps && cycles == 0) (void)hipMemsetAsync(ca.stamps, 0, 128, stream);
            ca.spread = lds_spread ? 1 : 0;
            ca.flag_a = (unsigned long long *)(part_spare + 512);   // 2 x 512 words
            ca.flag_b = ca.flag_a + kHoMaxWg;
            if (cyc_lds) (void)hipMemsetAsync(ca.flag_a, 0, 2 * kHoMaxWg * sizeof(unsigned long long), stream);
            const int lgrid = lds_spread ? kGmSub * gm.g : 8 * kGmSub * gm.g;
            if (cyc_lds && cyc_local)
                hipk_gm_solve_lds_kernel<T, true><<<lgrid, HIPK_THREADS, hipk_gm_solve_lds_bytes<T>(m), stream>>>(ca);
            else if (cyc_lds)
                hipk_gm_solve_lds_kernel<T, false><<<lgrid, HIPK_THREADS, hipk_gm_solve_lds_bytes<T>(m), stream>>>(ca);
            else
                hipk_gm_cycle_small_kernel<T><<<8 * gm.g, HIPK_BASE_CHUNK / hipk_vec<T>::VEC, 0, stream>>>(ca);
        }
        {
            if (mid_cycle) {
                hipk_gm_mid_args ca;
                memset(&ca, 0, sizeof(ca));
                ca.n = n;
                ca.g = gm.g;
                ca.win = mid_plan.max_slots * HIPK_TILE;
                ca.plan = mid_plan;
                ca.m = m;
                ca.crow = A->crow;
                ca.col = A->col;
                ca.val = A->val;
                ca.V = V;
                ca.ldv = ldv;
                ca.v_ll = (unsigned long long *)(vbase + (size_t)(m + 2) * vec);      // behind the basis and tmp (hipk_gmres_work_bytes)
                ca.slots = (unsigned long long *)(vbase + (size_t)(m + 2) * vec + hipk_align_up((size_t)n * 16, 256));
                ca.dinv = dinv;
                ca.scal = scal;
                ca.eps = eps_t;
                ca.slot_stride = 16;
                ca.xcd_aware = 1;
                {
                    const char *fe = getenv("HIPK_TEST_LDS_NOT_RESIDENT");
                    const int fail_launch = fe ? (atoi(fe) > 1 ? atoi(fe) : 1) : 0;
                    ca.test_not_resident = (++lds_launch_no == fail_launch) ? 1 : 0;
                }
                (void)hipMemsetAsync(ca.v_ll, 0, hipk_align_up((size_t)n * 16, 256), stream);
                (void)hipMemsetAsync(ca.slots, 0, (size_t)kGmMidKinds * gm.g * ca.slot_stride * 16, stream);
                mid_kern<<<hipk_xcd_grid(gm.g), 1024, mid_lds, stream>>>(ca);
            }
        }
        for (int k = (cyc || mid_cycle) ? m : k_start; k < m; ++k) {
            T *w = V + (int64_t)(k + 1) * ldv;
            hipk_spmv_args sw = sa;
            sw.x = V + (int64_t)k * ldv;
            sw.y = w;
            sw.skip_combine = small ? 1 : 0;  // small systems: hipk_gm_normalize_kernel folds the tile sums itself
            sw.mode = ext ? 0 : (HIPK_SPMV_DOT_YY | scale_bit);  // w = M(A v_k), ||w||^2 of the scaled vector (TSL:351-352)
            sw.part0 = part_spare;
            sw.part1 = part_ww;
            sw.stop_it = &scal->stop_step;
            sw.it = k;
            if ((rc = hipk_launch_spmv(A, sw, stream, &prof)) != HIPK_OK) break;
            if (ext && (rc = precondition(w, part_ww)) != HIPK_OK) break;  // w = M(A v_k), ||w||^2 (TSL:351-352)
            const int npass = (spec && !predict[k]) ? 1 : 2;
            for (int pass = 0; pass < npass; ++pass) {
                const dim3 mgrid(gm.g, k / 8 + 1);
                if (small) {  // 7 instead of 10 launches per Arnoldi step (decide and the two hreduce folded away)
                    if (wide)
                        hipk_gm_multidot_wide_kernel<T><<<mgrid, HIPK_BASE_CHUNK / hipk_vec<T>::VEC, 0, stream>>>(
                            n, scal, k, pass, V, ldv, w, part_md, part_qq, gm.g, eps_t);
                    else
                        hipk_gm_multidot_kernel<T, true><<<mgrid, HIPK_THREADS, 0, stream>>>(n, gm.ch, scal, k, pass, V, ldv, w,
                                                                                             part_md, part_qq, gm.g, eps_t);
                    if (wide)
                        hipk_gm_update_wide_kernel<T><<<gm.g, HIPK_BASE_CHUNK / hipk_vec<T>::VEC, 0, stream>>>(
                            n, scal, k, pass, V, ldv, w, part_qq, part_md, gm.g);
                    else
                        hipk_gm_update_kernel<T, true><<<gm.g, HIPK_THREADS, 0, stream>>>(n, gm.ch, scal, k, pass, V, ldv, w,
                                                                                          part_qq, part_md, gm.g);
                } else {
                    if (pass == 1) hipk_gm_decide_kernel<<<1, HIPK_THREADS, 0, stream>>>(scal, k, gm.g, part_qq, eps_t);
                    if (stream_k) {
#define HIPK_MD(NC, GW) hipk_gm_multidot_stream_kernel<T, NC><<<gm.g * (k / GW + 1), HIPK_THREADS, 0, stream>>>(n, gm.ch, scal, k, pass, V, ldv, w, part_md, gm.g, gm_nres)
                        if (k == 0) HIPK_MD(1, 8);
                        else if (k == 1) HIPK_MD(2, 8);
                        else if (k < 4) HIPK_MD(4, 8);
                        else if (k < 8 || !md_wide) HIPK_MD(8, 8);
                        else if (k < 16) HIPK_MD(16, 16);
                        else HIPK_MD(32, 32);
#undef HIPK_MD
                    }
                    else
                        hipk_gm_multidot_kernel<T, false><<<mgrid, HIPK_THREADS, 0, stream>>>(n, gm.ch, scal, k, pass, V, ldv, w,
                                                                                              part_md, part_qq, gm.g, eps_t);
                    hipk_gm_hreduce_kernel<<<k + 1, HIPK_THREADS, 0, stream>>>(scal, k, pass, gm.g, part_md);
                    if (stream_k && k < 8)
                        hipk_gm_update_stream_kernel<T, 8><<<gm.g, HIPK_THREADS, 0, stream>>>(n, gm.ch, scal, k, pass, V, ldv, w,
                                                                                               part_qq, gm_nres);
                    else if (stream_k) {
#define HIPK_UP(KC) hipk_gm_update_stream_kernel<T, KC><<<gm.g, HIPK_THREADS, 0, stream>>>(n, gm.ch, scal, k, pass, V, ldv, w, part_qq, gm_nres)
                        if (k < 32) HIPK_UP(32);
                        else HIPK_UP(256);
#undef HIPK_UP
                    } else
                        hipk_gm_update_kernel<T, false><<<gm.g, HIPK_THREADS, 0, stream>>>(n, gm.ch, scal, k, pass, V, ldv, w,
                                                                                           part_qq, part_md, gm.g);
                }
            }
            if (split_norm) {   // large systems: one workgroup for the scalars, a flat grid for the scaling (see hipk_gm_hcol_kernel)
                hipk_gm_hcol_kernel<<<1, HIPK_THREADS, 0, stream>>>(gm.g, scal, k, part_qq, part_ww, eps_t, npass == 1 ? 1 : 0,
                                                                   sizeof(T) == 4 ? 1 : 0);
                hipk_gm_scale_kernel<T><<<(unsigned)((n + HIPK_BASE_CHUNK - 1) / HIPK_BASE_CHUNK), HIPK_THREADS, 0, stream>>>(n, scal, k, w);
            } else
                hipk_gm_normalize_kernel<T><<<gm.g, HIPK_THREADS, 0, stream>>>(
                    n, gm.ch, gm.g, scal, k, w, part_qq, (small && !ext) ? A->tile_part + 4 * (size_t)nt : part_ww, eps_t,
                    (small && !ext) ? nt : 0, npass == 1 ? 1 : 0);
        }
        if (rc != HIPK_OK) break;
        if (hipGetLastError() != hipSuccess || hipMemcpyAsync(hs, scal, sizeof(*hs), hipMemcpyDeviceToHost, stream) != hipSuccess ||
            (big_n && hipMemcpyAsync(big_host.data(), big, big_n * sizeof(double), hipMemcpyDeviceToHost, stream) != hipSuccess) ||
            hipStreamSynchronize(stream) != hipSuccess) {
            hipk_set_error("hipk_gmres_solve: HIP failure inside a restart cycle");
            rc = HIPK_ERR_HIP;
            break;
        }
        if (hs->redo < 0 && mid_cycle) {
            // the workgroups of the one-launch step loop did not all arrive (nothing of the cycle is kept) or one of its hand-offs
            // never completed
            if (hs->redo == -3) {
                hipk_set_error("hipk_gmres_solve: a resident workgroup of the one-launch cycle stopped arriving");
                rc = HIPK_ERR_HIP;
                break;
            }
            if (!getenv("HIPK_TEST_LDS_NOT_RESIDENT")) mid_failed = true;
            mid_cycle = false;
            continue;
        }
        if (hs->redo < 0) {
            // the resident workgroups of a one-launch cycle did not all arrive (the device is shared and they were not
            // co-resident): nothing of the cycle is kept -- column 0 is untouched -- and this solve goes on with one launch
            // per kernel
            if (getenv("HIPK_GM_STAMPS")) fprintf(stderr, "hipk_gmres_solve: one-launch cycle abandoned (workgroups not co-resident)\n");
            if (hs->redo == -2 && cyc_lds && cyc_local) {
                cyc_local = false;      // its workgroups were spread over several XCDs: hand-offs at agent scope from now on
            } else {
                if (cyc_lds && !getenv("HIPK_TEST_LDS_NOT_RESIDENT")) lds_cycle_failed = true;
                cyc = cyc_lds = false;
            }
            continue;
        }
        if (cyc && cyc_lds) {  // hipk_gm_solve_lds_kernel ran whole cycles, loop test included
            cycles += hs->rep_cycles;
            matvecs += hs->rep_matvecs;
            if (hs->rep_breakdown) happy = 1;
            res_norm = hs->res_norm;
            if (hs->rep_status != 2) continue;
            // 2: the Arnoldi steps of one more cycle are in memory, its normal equations were not positive definite:
            // the general solve and the rest of that cycle below
        }
        if (hs->redo > 0) {  // speculation miss: second pass wanted at redo_step; enqueue the cycle again from there
            k_start = (int)hs->redo_step;
            predict[k_start] = true;
            hipk_gm_resume_kernel<<<1, 1, 0, stream>>>(scal);
            goto enqueue;
        }
        const int k = (int)hs->steps_done;
        matvecs += k;
        if (prof_valid == cycles * m) prof_valid += k;  // leading launches that did work
        if (hs->breakdown) happy = 1;
        hipk_gm_yN<HIPK_GM_MAXM_BIG + 1> yb;
        memset(&yb, 0, sizeof(yb));
        if (k > 0) {
            // the cycle's small arrays: the struct's own (ld 32) or the workspace block's host copy (hipk_gm_cycle_init_kernel's layout)
            const int ldh = big_n ? hipk_gm_big_ld(m) : HIPK_GM_LDH;
            const double *Hh = big_n ? big_host.data() : hs->H;
            const double *Rh = big_n ? Hh + (size_t)(m + 2) * ldh : hs->R;
            const double *bvh = big_n ? Rh + (size_t)ldh * ldh + 2 * ldh : hs->beta_vec;
            if (!incremental) {
                hipk_lstsq_normal(Hh, ldh, k, res_norm, yb.y);
            } else {
                for (int i = k - 1; i >= 0; --i) {  // solve_triangular, TSL:630
                    double s = bvh[i];
                    for (int p = i + 1; p < k; ++p) s = fma(-Rh[i * ldh + p], yb.y[p], s);
                    yb.y[i] = s / Rh[i * ldh + i];
                }
            }
            if (k <= HIPK_GM_LDH) {
                hipk_gm_y yy;
                memcpy(yy.y, yb.y, sizeof(yy.y));
                hipk_gm_xupdate_kernel<T><<<gm.g, HIPK_THREADS, 0, stream>>>(n, gm.ch, k, V, ldv, x, yy);
            } else {
                hipk_gm_xupdate_kernel<T, HIPK_GM_MAXM_BIG + 1><<<gm.g, HIPK_THREADS, 0, stream>>>(n, gm.ch, k, V, ldv, x, yb);
            }
        }
        if ((rc = hipk_launch_spmv(A, sr, stream)) != HIPK_OK) break;
        ++matvecs;
        if (ext && (rc = precondition(V, part_res)) != HIPK_OK) break;
        hipk_gm_resnorm_kernel<T><<<gm.g, HIPK_THREADS, 0, stream>>>(n, gm.ch, gm.g, scal, V, part_res, part_bb, eps_t);
        if (hipGetLastError() != hipSuccess || hipMemcpyAsync(head, scal, sizeof(head), hipMemcpyDeviceToHost, stream) != hipSuccess ||
            hipStreamSynchronize(stream) != hipSuccess) {
            hipk_set_error("hipk_gmres_solve: HIP failure at the end of a restart cycle");
            rc = HIPK_ERR_HIP;
            break;
        }
        res_norm = head[0];
        ++cycles;
    }
    free(hs);
    if (rc != HIPK_OK) return rc;
    if ((cyc || cyc_lds) && getenv("HIPK_GM_STAMPS")) {  // diagnostic build-in: where workgroup 0 of the cycle kernel spent its shader clocks
        unsigned long long st8[16];
        HIPK_CHECK_HIP(hipMemcpyAsync(st8, part_spare + 1600, sizeof(st8), hipMemcpyDeviceToHost, stream));
        HIPK_CHECK_HIP(hipStreamSynchronize(stream));
        if (cyc_lds)
            fprintf(stderr, "hipk_gm_solve_lds_kernel stamps (shader clocks of workgroup 0, thread 0; %lld cycles of %d steps): A SpMV %llu | "
                            "multi-dot (+ CGS2 decision of the step before a second pass) %llu | hand-off 1 %llu | B: loads + fold %llu | wait %llu | "
                            "update %llu | wait %llu | <q,q> %llu | hand-off 2 %llu | C: folds %llu | wait %llu | normalise + H column %llu | "
                            "wait %llu | end of cycle: least squares %llu | x update, residual, norm %llu\n",
                    (long long)cycles, m, st8[0], st8[1], st8[2], st8[8], st8[9], st8[10], st8[11], st8[3], st8[4], st8[12], st8[13],
                    st8[14], st8[5], st8[6], st8[7]);
        else
            fprintf(stderr, "hipk_gm_cycle_small_kernel stamps (shader clocks of workgroup 0, thread 0; %lld cycles of %d steps): A SpMV %llu | "
                            "multi-dot %llu | barrier 1 %llu | B fold + update %llu | barrier 2 %llu | C decide + normalise %llu | barrier 3 %llu\n",
                    (long long)cycles, m, st8[0], st8[1], st8[2], st8[3], st8[4], st8[5], st8[6]);
    }

    // TSL:766-773
    hipk_spmv_args sf = sr;
    sf.y = tmp;
    if ((rc = hipk_launch_spmv(A, sf, stream)) != HIPK_OK) return rc;
    ++matvecs;
    if (ext && (rc = precondition(tmp, part_res)) != HIPK_OK) return rc;
    if ((rc = hipk_launch_dot_parts(n, x, x, A->dtype, part_xx, stream)) != HIPK_OK) return rc;
    hipk_gm_final_kernel<<<1, HIPK_THREADS, 0, stream>>>(scal, gm.g, part_res, part_xx);
    HIPK_CHECK_HIP(hipGetLastError());
    double fin[4];
    HIPK_CHECK_HIP(hipEventRecord(whole.b, stream));
    HIPK_CHECK_HIP(hipMemcpyAsync(fin, scal, sizeof(fin), hipMemcpyDeviceToHost, stream));
    HIPK_CHECK_HIP(hipStreamSynchronize(stream));
    st->iterations = cycles;
    st->matvecs = matvecs;
    st->b_norm = b_norm;
    st->residual_norm = hipk_norm_from_sq(fin[2]);
    st->x_norm = hipk_norm_from_sq(fin[3]);
    st->threshold = atol_eff * 10;  // TSL:769
    st->info = (isnan(st->x_norm) || st->residual_norm > st->threshold) ? -1 : 0;
    st->breakdown = happy;
    st->recurrence_rs = res_norm;
    float ms = 0.f;
    HIPK_CHECK_HIP(hipEventElapsedTime(&ms, whole.a, whole.b));
    st->solve_ms = ms;
    HIPK_CHECK_HIP(prof.collect(st, prof_valid));
    return HIPK_OK;
}

extern "C" int hipk_gmres_solve(hipk_csr_t A, const void *b, void *x, void *work, size_t work_bytes,
                                const hipk_params *prm, hipk_stats *st, hipk_stream_t stream) {
    HIPK_REQUIRE(A && b && x && work && prm && st, HIPK_ERR_ARG, "null argument");
    HIPK_REQUIRE(A->n_rows == A->n_cols, HIPK_ERR_ARG, "linear operator must be a square matrix");
    HIPK_REQUIRE(A->n_rows > 0, HIPK_ERR_ARG, "empty system");
    HIPK_REQUIRE(prm->restart >= 1 && prm->restart <= HIPK_GM_MAXM_BIG, HIPK_ERR_UNSUPPORTED,
                 "restart must be in [1, 255] on the HIP path");
    HIPK_REQUIRE(prm->gmres_method == HIPK_GMRES_BATCHED || prm->gmres_method == HIPK_GMRES_INCREMENTAL, HIPK_ERR_ARG,
                 "Unsupported solve_method");
    HIPK_REQUIRE(hipk_aligned16(b) && hipk_aligned16(x) && (((uintptr_t)work) & 255u) == 0, HIPK_ERR_ALIGN,
                 "b/x must be 16-byte and work 256-byte aligned");
    HIPK_REQUIRE(work_bytes >= hipk_gmres_work_bytes(A->n_rows, prm->restart, A->dtype), HIPK_ERR_WORKSPACE,
                 "work too small");
    HIPK_REQUIRE(b != x, HIPK_ERR_ARG, "b and x must not alias");
    memset(st, 0, sizeof(*st));
    if (A->dtype == HIPK_F64)
        return hipk_gmres_solve_t<double>(A, nullptr, (const double *)b, (double *)x, (char *)work, prm, st,
                                          (hipStream_t)stream);
    return hipk_gmres_solve_t<float>(A, nullptr, (const float *)b, (float *)x, (char *)work, prm, st, (hipStream_t)stream);
}

extern "C" int hipk_pgmres_solve(hipk_csr_t A, const void *dinv, const void *b, void *x, void *work, size_t work_bytes,
                                 const hipk_params *prm, hipk_stats *st, hipk_stream_t stream) {
    HIPK_REQUIRE(A && dinv && b && x && work && prm && st, HIPK_ERR_ARG, "null argument");
    HIPK_REQUIRE(A->n_rows == A->n_cols, HIPK_ERR_ARG, "linear operator must be a square matrix");
    HIPK_REQUIRE(A->n_rows > 0, HIPK_ERR_ARG, "empty system");
    HIPK_REQUIRE(prm->restart >= 1 && prm->restart <= HIPK_GM_MAXM_BIG, HIPK_ERR_UNSUPPORTED,
                 "restart must be in [1, 255] on the HIP path");
    HIPK_REQUIRE(prm->gmres_method == HIPK_GMRES_BATCHED || prm->gmres_method == HIPK_GMRES_INCREMENTAL, HIPK_ERR_ARG,
                 "Unsupported solve_method");
    HIPK_REQUIRE(hipk_aligned16(b) && hipk_aligned16(x) && hipk_aligned16(dinv) && (((uintptr_t)work) & 255u) == 0,
                 HIPK_ERR_ALIGN, "b/x/dinv must be 16-byte and work 256-byte aligned");
    HIPK_REQUIRE(work_bytes >= hipk_gmres_work_bytes(A->n_rows, prm->restart, A->dtype), HIPK_ERR_WORKSPACE,
                 "work too small");
    HIPK_REQUIRE(b != x, HIPK_ERR_ARG, "b and x must not alias");
    memset(st, 0, sizeof(*st));
    if (A->dtype == HIPK_F64)
        return hipk_gmres_solve_t<double>(A, (const double *)dinv, (const double *)b, (double *)x, (char *)work, prm, st,
                                          (hipStream_t)stream);
    return hipk_gmres_solve_t<float>(A, (const float *)dinv, (const float *)b, (float *)x, (char *)work, prm, st,
                                     (hipStream_t)stream);
}

extern "C" int hipk_pgmres_solve_cb(hipk_csr_t A, hipk_precond_fn M, void *user, const void *b, void *x, void *work,
                                    size_t work_bytes, const hipk_params *prm, hipk_stats *st, hipk_stream_t stream) {
    HIPK_REQUIRE(A && M && b && x && work && prm && st, HIPK_ERR_ARG, "null argument");
    HIPK_REQUIRE(A->n_rows == A->n_cols, HIPK_ERR_ARG, "linear operator must be a square matrix");
    HIPK_REQUIRE(A->n_rows > 0, HIPK_ERR_ARG, "empty system");
    HIPK_REQUIRE(prm->restart >= 1 && prm->restart <= HIPK_GM_MAXM_BIG, HIPK_ERR_UNSUPPORTED,
                 "restart must be in [1, 255] on the HIP path");
    HIPK_REQUIRE(prm->gmres_method == HIPK_GMRES_BATCHED || prm->gmres_method == HIPK_GMRES_INCREMENTAL, HIPK_ERR_ARG,
                 "Unsupported solve_method");
    HIPK_REQUIRE(hipk_aligned16(b) && hipk_aligned16(x) && (((uintptr_t)work) & 255u) == 0, HIPK_ERR_ALIGN,
                 "b/x must be 16-byte and work 256-byte aligned");
    HIPK_REQUIRE(work_bytes >= hipk_gmres_work_bytes(A->n_rows, prm->restart, A->dtype), HIPK_ERR_WORKSPACE,
                 "work too small");
    HIPK_REQUIRE(b != x, HIPK_ERR_ARG, "b and x must not alias");
    memset(st, 0, sizeof(*st));
    if (A->dtype == HIPK_F64)
        return hipk_gmres_solve_t<double>(A, nullptr, (const double *)b, (double *)x, (char *)work, prm, st,
                                          (hipStream_t)stream, M, user);
    return hipk_gmres_solve_t<float>(A, nullptr, (const float *)b, (float *)x, (char *)work, prm, st, (hipStream_t)stream, M,
                                     user);
}


// =====================================================================================================================
// Row-partitioned GMRES, the loop of one rank in C (new against the reference, which is single-device).  The algorithm is `gmres`
// (TSL:641-803: `_gmres_batched` / `_gmres_incremental`) on the kernels of the large-system path above; each of them folds the
// chunk partials of ALL ranks, so the iterates, H, the cycle and operator-application counts are bitwise those of the single-GPU
// solve for any rank count.  A rank's kernels write their chunk partials at the rank's position of the GLOBAL partial arrays and
// the ranks complete them with in-place all-gathers; per Arnoldi step on the solver's stream:
//   halo of v_k | SpMV, all-gather ||w||^2 | multi-dot, all-gather of its k+1 columns (one group) | h-reduce | update, all-gather ||q||^2
//   | [CGS2 decision, second pass: the same three again -- device no-ops when not wanted, the collectives still pair up] | normalise
// The cycle's small least-squares problem is solved by every rank's host from its own (identical) copy of H.
// Conventions of hipk_dist_cg_solve (csrc/hipk_dist.hip): same plan / collective structs, fp64, M = identity.
static size_t hipk_dgm_vec_bytes(const hipk_dist_plan *pl) {
    return hipk_align_up((size_t)(pl->n_ext > 0 ? pl->n_ext : 1) * sizeof(double), 256);
}
extern "C" size_t hipk_dist_gmres_work_bytes(const hipk_dist_plan *plan, int restart) {
    if (!plan || plan->world < 1 || plan->per < 1) return 0;
    const int m = restart < 1 ? 1 : (restart > HIPK_GM_MAXM ? HIPK_GM_MAXM : restart);
    const size_t slab = (size_t)(plan->slab > 0 ? plan->slab : 1);
    return kGmHeader + (size_t)(kGmSlots + m + 1) * HIPK_MAX_PARTS * sizeof(double) + (size_t)(m + 2) * hipk_dgm_vec_bytes(plan) +
           hipk_align_up((size_t)(plan->n_send > 0 ? plan->n_send : 1) * 8, 256) + hipk_align_up(slab * 8, 256) +
           hipk_align_up(slab * (size_t)plan->world * 8, 256);
}

extern "C" int hipk_dist_gmres_solve(hipk_csr_t A, const hipk_dist_plan *pl, const hipk_rccl *cc, const void *b_local, void *x_ext,
                                     void *work_, size_t work_bytes, const hipk_params *prm, hipk_stats *st, hipk_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    HIPK_REQUIRE(A && pl && cc && b_local && x_ext && work_ && prm && st, HIPK_ERR_ARG, "null argument");
    HIPK_REQUIRE(A->dtype == HIPK_F64, HIPK_ERR_UNSUPPORTED, "the row-partitioned solver is fp64");
    HIPK_REQUIRE(prm->restart >= 1 && prm->restart <= HIPK_GM_MAXM, HIPK_ERR_ARG, "restart must be in [1, 31]");
    HIPK_REQUIRE(pl->world >= 1 && pl->rank >= 0 && pl->rank < pl->world, HIPK_ERR_ARG, "rank / world");
    HIPK_REQUIRE(pl->n_local > 0 && pl->n_local == A->n_rows && pl->n_ext >= pl->n_local, HIPK_ERR_ARG,
                 "every rank must own rows (n_local > 0) and n_ext >= n_local");
    HIPK_REQUIRE(pl->per >= 1 && (int64_t)pl->per * pl->world >= pl->g_red && pl->g_red >= 1 && pl->g_red <= HIPK_MAX_PARTS &&
                     (int64_t)pl->per * pl->world <= HIPK_MAX_PARTS,
                 HIPK_ERR_ARG, "partial-sum geometry (per * world must not exceed 2048)");
    HIPK_REQUIRE((pl->n_local + pl->chunk_rows - 1) / pl->chunk_rows <= pl->per, HIPK_ERR_ARG, "more local chunks than `per`");
    HIPK_REQUIRE(cc->all_gather && cc->group_start && cc->group_end && (pl->world == 1 || pl->halo_mode == 0 || (cc->send && cc->recv)),
                 HIPK_ERR_ARG, "missing collective entry points");
    HIPK_REQUIRE((((uintptr_t)work_) & 255u) == 0 && hipk_aligned16(x_ext) && hipk_aligned16(b_local), HIPK_ERR_ALIGN,
                 "work must be 256-byte, x / b 16-byte aligned");
    HIPK_REQUIRE(work_bytes >= hipk_dist_gmres_work_bytes(pl, prm->restart), HIPK_ERR_WORKSPACE, "work too small");
    memset(st, 0, sizeof(*st));
    typedef double T;
    char *work = (char *)work_;
    const int64_t n = pl->n_local, n_ext = pl->n_ext;
    const int ch = pl->chunk_rows, G = pl->g_red, per = pl->per, W = pl->world;
    const int gl = (int)((n + ch - 1) / ch);             // local chunks = grid of the vector kernels
    const int c0 = pl->rank * per;                       // this rank's position in the global partial arrays
    const int m = prm->restart;
    const size_t vec = hipk_dgm_vec_bytes(pl);
    const int64_t ldv = (int64_t)(vec / sizeof(T));
    hipk_gm_scal *scal = (hipk_gm_scal *)work;
    double *parts = (double *)(work + kGmHeader);
    double *part_ww = parts, *part_qq = parts + HIPK_MAX_PARTS, *part_res = parts + 2 * HIPK_MAX_PARTS;
    double *part_bb = parts + 3 * HIPK_MAX_PARTS, *part_xx = parts + 4 * HIPK_MAX_PARTS;
    double *part_spare = parts + 5 * HIPK_MAX_PARTS;
    double *part_md = parts + (size_t)kGmSlots * HIPK_MAX_PARTS;
    char *vbase = work + kGmHeader + (size_t)(kGmSlots + m + 1) * HIPK_MAX_PARTS * sizeof(double);
    T *V = (T *)vbase;
    T *tmp = (T *)(vbase + (size_t)(m + 1) * vec);
    char *cbase = vbase + (size_t)(m + 2) * vec;
    double *send_buf = (double *)cbase;
    double *slab_loc = (double *)(cbase + hipk_align_up((size_t)(pl->n_send > 0 ? pl->n_send : 1) * 8, 256));
    double *slab_all = (double *)((char *)slab_loc + hipk_align_up((size_t)(pl->slab > 0 ? pl->slab : 1) * 8, 256));
    T *x = (T *)x_ext;
    const T *b = (const T *)b_local;
    const int incremental = (prm->gmres_method == HIPK_GMRES_INCREMENTAL) ? 1 : 0;
    const double eps_t = HIPK_EPS64;
    const int64_t maxiter = (prm->maxiter < 0) ? 10 * pl->n_global : prm->maxiter;  // TSL:719-721
    const int NCCL_F64 = 8;
    enum { MODE_DOT_YY = 2, MODE_RESID = 4 };

#define HIPK_DGM_NCCL(expr, what)                                                       \
    do {                                                                                \
        const int _r = (expr);                                                          \
        if (_r != 0) {                                                                  \
            hipk_set_error("hipk_dist_gmres_solve: %s failed (ncclResult %d)", what, _r); \
            return HIPK_ERR_HIP;                                                        \
        }                                                                               \
    } while (0)
#define HIPK_DGM_TRY(expr)              \
    do {                                \
        const int _rc = (expr);         \
        if (_rc != HIPK_OK) return _rc; \
    } while (0)

    hipk_event_pair whole;
    HIPK_CHECK_HIP(whole.create());
    HIPK_CHECK_HIP(hipEventRecord(whole.a, stream));
    HIPK_CHECK_HIP(hipMemsetAsync(work, 0, hipk_dist_gmres_work_bytes(pl, m), stream));
    if (n_ext > n) HIPK_CHECK_HIP(hipMemsetAsync(x + n, 0, (size_t)(n_ext - n) * 8, stream));

    // in-place all-gather of a global partial array: every rank wrote its `per` entries at arr + c0
    auto complete = [&](double *arr) -> int {
        HIPK_DGM_NCCL(cc->all_gather(arr + c0, arr, (size_t)per, NCCL_F64, cc->comm, stream), "all_gather(partials)");
        return HIPK_OK;
    };
    bool need_pack = false;
    for (int peer = 0; peer < W; ++peer)
        if (pl->send_counts[peer] > 0 && !(pl->send_first && pl->send_first[peer] >= 0)) need_pack = true;
    auto halo = [&](double *v) -> int {   // the peers' entries this rank's rows reference -> v[n .. n_ext)
        if (W == 1 || (pl->n_send == 0 && pl->n_ghost == 0 && pl->halo_mode == 1)) return HIPK_OK;
        if (pl->halo_mode == 1) {
            if (pl->n_send && need_pack) HIPK_DGM_TRY(hipk_gather(pl->n_send, pl->send_idx_dev, v, send_buf, HIPK_F64, stream));
            HIPK_DGM_NCCL(cc->group_start(), "group_start");
            size_t so = 0, ro = 0;
            for (int peer = 0; peer < W; ++peer) {
                const size_t ns = (size_t)pl->send_counts[peer], nr = (size_t)pl->recv_counts[peer];
                const bool direct = pl->send_first && pl->send_first[peer] >= 0;
                if (ns) HIPK_DGM_NCCL(cc->send(direct ? v + pl->send_first[peer] : send_buf + so, ns, NCCL_F64, peer, cc->comm, stream), "send(halo)");
                if (nr) HIPK_DGM_NCCL(cc->recv(v + n + ro, nr, NCCL_F64, peer, cc->comm, stream), "recv(halo)");
                so += ns;
                ro += nr;
            }
            HIPK_DGM_NCCL(cc->group_end(), "group_end");
        } else {
            if (pl->n_send) HIPK_DGM_TRY(hipk_gather(pl->n_send, pl->send_idx_dev, v, slab_loc, HIPK_F64, stream));
            HIPK_DGM_NCCL(cc->all_gather(slab_loc, slab_all, (size_t)pl->slab, NCCL_F64, cc->comm, stream), "all_gather(halo slabs)");
            if (pl->n_ghost) HIPK_DGM_TRY(hipk_gather(pl->n_ghost, pl->ghost_src_dev, slab_all, v + n, HIPK_F64, stream));
        }
        return HIPK_OK;
    };

    // residual = b - A x0 into column 0, unit residual + norm (TSL:791-792); <b,b>
    auto residual = [&]() -> int {
        HIPK_DGM_TRY(halo(x));
        HIPK_DGM_TRY(hipk_spmv_ex(A, x, V, MODE_RESID | MODE_DOT_YY, nullptr, b, part_spare + c0, part_res + c0, nullptr, 0, stream));
        HIPK_DGM_TRY(complete(part_res));
        hipk_gm_resnorm_kernel<T><<<gl, HIPK_THREADS, 0, stream>>>(n, ch, G, scal, V, part_res, part_bb, eps_t);
        return hipGetLastError() == hipSuccess ? HIPK_OK : HIPK_ERR_HIP;
    };
    HIPK_DGM_TRY(hipk_dot_parts(n, ch, b, b, HIPK_F64, part_bb + c0, stream));
    HIPK_DGM_TRY(complete(part_bb));
    HIPK_DGM_TRY(residual());
    int64_t matvecs = 1;
    double head[2];
    HIPK_CHECK_HIP(hipMemcpyAsync(head, scal, sizeof(head), hipMemcpyDeviceToHost, stream));
    HIPK_CHECK_HIP(hipStreamSynchronize(stream));
    double res_norm = head[0];
    const double bs = head[1];
    const double b_norm = hipk_norm_from_sq(bs);
    // TSL:735-753 on the GLOBAL size
    const double eps = HIPK_EPS64;
    const double ng = (double)pl->n_global;
    const double cand = (prm->gpu_tolerances ? 1e-12 : 1e-14) * sqrt(ng);
    const double adaptive = (cand > prm->tol) ? cand : (double)(float)prm->tol;
    const double base_atol = (double)(float)(eps * (prm->gpu_tolerances ? 1000 : 100) * ng);
    const double atol_eff = hipk_tmax(adaptive * b_norm, hipk_tmax((double)(float)prm->atol, base_atol));
    const double ptol = b_norm * hipk_tmin(1.0, atol_eff / b_norm);

    std::vector<unsigned char> hs_store(sizeof(hipk_gm_scal));
    hipk_gm_scal *hs = (hipk_gm_scal *)hs_store.data();
    const int nres = 5;
    int64_t cycles = 0;
    int happy = 0;
    while (cycles < maxiter && res_norm > atol_eff) {
        hipk_gm_cycle_init_kernel<<<1, HIPK_THREADS, 0, stream>>>(scal, incremental, ptol);
        for (int k = 0; k < m; ++k) {
            T *vk = V + (int64_t)k * ldv, *w = V + (int64_t)(k + 1) * ldv;
            HIPK_DGM_TRY(halo(vk));
            HIPK_DGM_TRY(hipk_spmv_ex(A, vk, w, MODE_DOT_YY, nullptr, nullptr, part_spare + c0, part_ww + c0, &scal->stop_step, k, stream));
            HIPK_DGM_TRY(complete(part_ww));
            for (int pass = 0; pass < 2; ++pass) {
                if (pass == 1) hipk_gm_decide_kernel<<<1, HIPK_THREADS, 0, stream>>>(scal, k, G, part_qq, eps_t);
                const int mg = gl * (k / 8 + 1);
                hipk_gm_multidot_stream_kernel<T, 8><<<mg, HIPK_THREADS, 0, stream>>>(n, ch, scal, k, pass, V, ldv, w, part_md + c0, gl, nres);
                if (W > 1) HIPK_DGM_NCCL(cc->group_start(), "group_start");
                for (int j = 0; j <= k; ++j) HIPK_DGM_TRY(complete(part_md + (size_t)j * HIPK_MAX_PARTS));
                if (W > 1) HIPK_DGM_NCCL(cc->group_end(), "group_end");
                hipk_gm_hreduce_kernel<<<k + 1, HIPK_THREADS, 0, stream>>>(scal, k, pass, G, part_md);
                hipk_gm_update_stream_kernel<T, HIPK_GM_LDH><<<gl, HIPK_THREADS, 0, stream>>>(n, ch, scal, k, pass, V, ldv, w, part_qq + c0, nres);
                HIPK_DGM_TRY(complete(part_qq));
            }
            hipk_gm_normalize_kernel<T><<<gl, HIPK_THREADS, 0, stream>>>(n, ch, G, scal, k, w, part_qq, part_ww, eps_t, 0, 0);
        }
        if (hipGetLastError() != hipSuccess || hipMemcpyAsync(hs, scal, sizeof(*hs), hipMemcpyDeviceToHost, stream) != hipSuccess ||
            hipStreamSynchronize(stream) != hipSuccess) {
            hipk_set_error("hipk_dist_gmres_solve: HIP failure inside a restart cycle");
            return HIPK_ERR_HIP;
        }
        const int k = (int)hs->steps_done;
        matvecs += k;
        if (hs->breakdown) happy = 1;
        hipk_gm_y yy;
        memset(&yy, 0, sizeof(yy));
        if (k > 0) {
            if (!incremental) {
                hipk_lstsq_normal(hs->H, HIPK_GM_LDH, k, res_norm, yy.y);
            } else {
                for (int i = k - 1; i >= 0; --i) {  // solve_triangular, TSL:630
                    double sacc = hs->beta_vec[i];
                    for (int p = i + 1; p < k; ++p) sacc = fma(-hs->R[i * HIPK_GM_LDH + p], yy.y[p], sacc);
                    yy.y[i] = sacc / hs->R[i * HIPK_GM_LDH + i];
                }
            }
            hipk_gm_xupdate_kernel<T><<<gl, HIPK_THREADS, 0, stream>>>(n, ch, k, V, ldv, x, yy);
        }
        HIPK_DGM_TRY(residual());
        ++matvecs;
        if (hipMemcpyAsync(head, scal, sizeof(head), hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) {
            hipk_set_error("hipk_dist_gmres_solve: HIP failure at the end of a restart cycle");
            return HIPK_ERR_HIP;
        }
        res_norm = head[0];
        ++cycles;
    }

    // TSL:766-773
    HIPK_DGM_TRY(halo(x));
    HIPK_DGM_TRY(hipk_spmv_ex(A, x, tmp, MODE_RESID | MODE_DOT_YY, nullptr, b, part_spare + c0, part_res + c0, nullptr, 0, stream));
    HIPK_DGM_TRY(complete(part_res));
    ++matvecs;
    HIPK_DGM_TRY(hipk_dot_parts(n, ch, x, x, HIPK_F64, part_xx + c0, stream));
    HIPK_DGM_TRY(complete(part_xx));
    hipk_gm_final_kernel<<<1, HIPK_THREADS, 0, stream>>>(scal, G, part_res, part_xx);
    HIPK_CHECK_HIP(hipGetLastError());
    double fin[4];
    HIPK_CHECK_HIP(hipEventRecord(whole.b, stream));
    HIPK_CHECK_HIP(hipMemcpyAsync(fin, scal, sizeof(fin), hipMemcpyDeviceToHost, stream));
    HIPK_CHECK_HIP(hipStreamSynchronize(stream));
    st->iterations = cycles;
    st->matvecs = matvecs;
    st->b_norm = b_norm;
    st->residual_norm = hipk_norm_from_sq(fin[2]);
    st->x_norm = hipk_norm_from_sq(fin[3]);
    st->threshold = atol_eff * 10;  // TSL:769
    st->info = (isnan(st->x_norm) || st->residual_norm > st->threshold) ? -1 : 0;
    st->breakdown = happy;
    st->recurrence_rs = res_norm;
    float ms = 0.f;
    HIPK_CHECK_HIP(hipEventElapsedTime(&ms, whole.a, whole.b));
    st->solve_ms = ms;
#undef HIPK_DGM_NCCL
#undef HIPK_DGM_TRY
    return HIPK_OK;
}

#ifdef HIPK_GM_STAMPS
// diagnostic twin only: per-workgroup phase time sums of the last hipk_gm_mid_kernel launch (hipk_gm_mid.h)
extern "C" int hipk_debug_gm_mid_stamps(unsigned long long *out, size_t count) {
    const size_t have = sizeof(hipk_gm_mid_stamps) / sizeof(unsigned long long);
    HIPK_CHECK_HIP(hipDeviceSynchronize());
    HIPK_CHECK_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(hipk_gm_mid_stamps), sizeof(unsigned long long) * (count < have ? count : have)));
    return HIPK_OK;
}
#endif
