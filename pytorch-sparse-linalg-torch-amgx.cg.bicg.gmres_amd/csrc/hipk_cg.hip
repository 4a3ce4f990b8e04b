// hipk_cg.hip -- device-resident conjugate gradient.
//
// Restates `_isolve(_cg_solve)` (TSL:806-856, 968-1016) for M = identity as three
// kernels per iteration, all scalars living in device memory:
//   K1 spmv+dot   Ap = A p, partials of <p,Ap>            (TSL:845-846)   B_spmv
//   K2 update     alpha = gamma/<p,Ap>; r -= alpha Ap; partials of <r,r>
//                                                          (TSL:846, 848-850)   24 n bytes
//   K3 direction  alpha again (same partials, same bits), beta = <r,r>/gamma; x += alpha p; p = r + beta p;
//                 gamma <- <r,r>; stop test                (TSL:847, 851-853, 841) 40 n bytes
// = B_spmv + 64 n bytes per iteration: the x update rides on the pass that already streams p, 8 n bytes
// less than the 72 n of SURVEY 8d; the arithmetic per element is unchanged.  Every workgroup re-derives alpha/beta from the chunk
// partials of the previous kernel with the fixed tree, so no grid barrier, no atomics
// and no host round trip are needed; the host follows the loop through a pinned word the direction kernel
// stores to (hipk_pacer, hipk_solve.h) and the kernels of iterations >= stop_it return immediately, so the
// solve stops at exactly the iteration the reference stops at.
#include <type_traits>
#include <math.h>
#include <stdlib.h>

#include <vector>

#include "hipk_blas1.h"
#include "hipk_solve.h"
#include "hipk_spmv.h"
#include "hipk_handoff.h"
#include "hipk_fx.h"

// progress / placement block of the one-launch loops (hipk_cg_solve_lds_kernel), zeroed before each launch
struct hipk_lds_ctl {
    double rs_last;    // <r,r> of the last finished iteration
    int64_t it_done;   // iterations finished when the launch returned
    int32_t redo;      // < 0: its resident workgroups did not all arrive / were spread over several XCDs (nothing was modified)
    int32_t bar;       // counter barrier of its placement check
    unsigned xcc_mask;
    unsigned pad;
};
struct hipk_cg_scal {
    double gamma[2];   // <r,r> ping-pong by iteration parity
    double atol2;      // max(tol^2 <b,b>, atol^2)            (TSL:815-817)
    double bs;         // <b,b>
    double res2;       // true ||b - A x||^2 after the loop  (TSL:1008)
    double xx;         // <x,x>                               (TSL:1013)
    int64_t stop_it;   // iterations >= stop_it are no-ops
    int64_t *host_sig; // pinned host word the direction kernel reports to (hipk_pacer), or null
    double dir_alpha;  // hipk_cg_scalars_kernel -> hipk_cg_direction_flat_kernel (streaming policy): gamma / <p,Ap>, <r,r> / gamma
    double dir_beta;
    hipk_lds_ctl ctl;  // hipk_cg_solve_lds_kernel (small systems: the whole loop in one launch)
};
static_assert(sizeof(hipk_cg_scal) <= 256, "the scalar block is 256 bytes");
#include "hipk_cg_mid.h"   // one-launch loop for mid-size systems (uses hipk_lds_ctl)

// gamma0 = <r0,r0>, bs = <b,b>, atol2; p = r0.
template <typename T>
__global__ __launch_bounds__(HIPK_THREADS) void hipk_cg_start_kernel(
    int64_t n, int ch, int g, hipk_cg_scal *__restrict__ scal, const double *__restrict__ part_rr,
    const double *__restrict__ part_bb, const T *__restrict__ r, T *__restrict__ p, double tol2, double atol_sq,
    int64_t maxiter, int64_t *host_sig = nullptr) {
    __shared__ double sbuf[2 * HIPK_THREADS];
    double gamma0, bs;
    hipk_reduce_parts2(part_rr, part_bb, g, gamma0, bs, sbuf);  // g = partial count of ALL ranks
    const int c = blockIdx.x;
    hipk_chunk_loop<T>(n, ch, c, [&](int64_t i, int nv) {
        T rv[hipk_vec<T>::VEC];
        hipk_ld<T>(r, i, nv, rv);
        hipk_st<T>(p, i, nv, rv);
    });
    if (c == 0 && threadIdx.x == 0) {
        const double a2 = tol2 * bs;
        const double atol2 = (a2 > atol_sq) ? a2 : atol_sq;  // torch.maximum: NaN-propagation irrelevant here
        scal->gamma[0] = gamma0;
        scal->gamma[1] = 0.0;
        scal->atol2 = atol2;
        scal->bs = bs;
        // TSL:841: `if k >= maxiter or rs <= atol2: break` evaluated before the first SpMV
        const bool done = (maxiter <= 0 || gamma0 <= atol2);
        scal->stop_it = done ? 0 : INT64_MAX;
        scal->host_sig = host_sig;
        if (done) hipk_signal(host_sig, HIPK_SIG_STOP);
    }
}

// The two per-iteration vector kernels run as ONE wave of workgroups (a chunk each, <= 2048 of them on 2048
// slots), so whatever a workgroup does before its first vector load is exposed in full: the stop word, the
// partial sums and the fold's barriers.  Both kernels therefore request the first HIPK_BASE_CHUNK elements of
// two operands BEFORE reading the stop word and folding the partials (hipk_pre, hipk_blas1.h; no store happens
// until the stop test has passed).  Order of operations per element is unchanged.
// SMALL (systems of <= 8 reduction chunks, launch-bound): <p,Ap> is folded here from the SpMV's per-wavefront tile
// sums (hipk_fold_tiles8: part_pAp then points at them, `ntiles` tiles), the combine launch is skipped.
// NT (systems whose CG working set -- x, r, p, Ap -- is far beyond the Infinity Cache): every vector is a stream.
template <typename T, bool SMALL = false, bool NT = false>
__global__ __launch_bounds__(HIPK_THREADS) void hipk_cg_update_kernel(
    int64_t n, int ch, int g, const hipk_cg_scal *__restrict__ scal, int64_t it,
    const double *__restrict__ part_pAp, const T *__restrict__ Ap, T *__restrict__ r, double *__restrict__ part_rr,
    int ntiles = 0) {
    const int c = blockIdx.x;
    hipk_pre<T, 2, NT> pre;
    pre.issue(n, ch, c, {Ap, (const T *)r});
    if (it >= scal->stop_it) return;
    __shared__ double sbuf[HIPK_THREADS];
    const double pAp = SMALL ? hipk_fold_tiles8(part_pAp, ntiles, ch / HIPK_TILE, g, sbuf)
                             : hipk_reduce_parts(part_pAp, g, sbuf);
    const double gamma = scal->gamma[it & 1];
    const T alpha = (T)(gamma / pAp);  // TSL:846
    double acc = 0.0;
    pre.run([&](int64_t i, int nv, T(&v)[2][hipk_vec<T>::VEC]) {
        constexpr int VEC = hipk_vec<T>::VEC;
        T rv[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            const T m1 = alpha * v[0][k];
            rv[k] = v[1][k] - m1;  // TSL:848
            if (k < nv) acc = fma((double)rv[k], (double)rv[k], acc);  // TSL:850
        }
        hipk_st<T>(r, i, nv, rv);
    });
    acc = hipk_block_sum(acc, sbuf);
    if (threadIdx.x == 0) part_rr[c] = acc;
}

// NOX: p only -- the row-partitioned solver with the x update on a side stream (hipk_cg_xupdate_kernel).  A compile-time switch:
// as a run-time branch it cost the hot instantiation its 8 workgroups per CU (74 VGPRs: 22.6 -> 25.4 us at N = 4 M).
template <typename T, bool SMALL = false, bool NT = false, bool NOX = false>
__global__ __launch_bounds__(HIPK_THREADS) void hipk_cg_direction_kernel(
    int64_t n, int ch, int g, hipk_cg_scal *__restrict__ scal, int64_t it, int64_t maxiter,
    const double *__restrict__ part_pAp, const double *__restrict__ part_rr, const T *__restrict__ r,
    T *__restrict__ p, T *__restrict__ x, int ntiles = 0) {
    const int c = blockIdx.x;
    // r and p are requested up front; x (needed last) is loaded step by step after the fold: all three would
    // take 76 VGPRs and drop the kernel to 6 workgroups per CU (1536 slots < 1954 chunks: a second round)
    // streaming policy (NT): x is the third batched operand, two steps per batch (instead of a load inside the step, behind the
    // previous step's stores); same-box A/B at N = 64 M: 457 vs 464 us, no gain (profiles/r02_vector_tail_ab.txt)
    typename std::conditional<NT && !NOX, hipk_pre<T, 3, true, 2>, hipk_pre<T, 2, NT>>::type pre;
    if constexpr (NT && !NOX) pre.issue(n, ch, c, {r, (const T *)p, (const T *)x});
    else pre.issue(n, ch, c, {r, (const T *)p});
    if (it >= scal->stop_it) return;
    __shared__ double sbuf[2 * HIPK_THREADS];
    double pAp, rr;
    if (SMALL) {
        pAp = hipk_fold_tiles8(part_pAp, ntiles, ch / HIPK_TILE, g, sbuf);
        rr = hipk_reduce_parts(part_rr, g, sbuf);
    } else {
        hipk_reduce_parts2(part_pAp, part_rr, g, pAp, rr, sbuf);
    }
    const double gamma = scal->gamma[it & 1];
    const T alpha = (T)(gamma / pAp);  // TSL:846, the same bits hipk_cg_update_kernel derived
    const T beta = (T)(rr / gamma);    // TSL:851
    if constexpr (NOX) {
        pre.run([&](int64_t i, int nv, T(&v)[2][hipk_vec<T>::VEC]) {
            constexpr int VEC = hipk_vec<T>::VEC;
            T pv[VEC];
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                const T m = beta * v[1][k];
                pv[k] = v[0][k] + m;  // TSL:852
            }
            hipk_st<T>(p, i, nv, pv);
        });
    } else if constexpr (NT) {
        pre.run([&](int64_t i, int nv, T(&v)[3][hipk_vec<T>::VEC]) {
            constexpr int VEC = hipk_vec<T>::VEC;
            T xv[VEC], pv[VEC];
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                const T m0 = alpha * v[1][k];
                xv[k] = v[2][k] + m0;  // TSL:847 (with the p of this iteration, before it is replaced)
                const T m = beta * v[1][k];
                pv[k] = v[0][k] + m;  // TSL:852
            }
            hipk_st_nt_vec<T>(x, i, nv, xv);  // x is not read again before the next direction kernel
            hipk_st<T>(p, i, nv, pv);
        });
    } else {
        pre.run([&](int64_t i, int nv, T(&v)[2][hipk_vec<T>::VEC]) {
            constexpr int VEC = hipk_vec<T>::VEC;
            T xv[VEC], pv[VEC];
            hipk_ld<T>((const T *)x, i, nv, xv);
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                const T m0 = alpha * v[1][k];
                xv[k] = xv[k] + m0;  // TSL:847 (with the p of this iteration, before it is replaced)
                const T m = beta * v[1][k];
                pv[k] = v[0][k] + m;  // TSL:852
            }
            hipk_st<T>(x, i, nv, xv);
            hipk_st<T>(p, i, nv, pv);
        });
    }
    if (c == 0 && threadIdx.x == 0) {
        scal->gamma[(it + 1) & 1] = rr;  // TSL:853
        // TSL:841 for the NEXT pass: stop when k+1 >= maxiter or rs <= atol2.
        // Workgroups of THIS launch compare against `it`, so they are unaffected.
        const bool done = (it + 1 >= maxiter || rr <= scal->atol2);
        if (done) scal->stop_it = it + 1;
        hipk_signal(scal->host_sig, done ? (HIPK_SIG_STOP | (it + 1)) : (it + 1));
    }
}

// ---- TWO launches per iteration for launch-bound mid-size systems (33 .. 150 reduction chunks: 65 k < n <= 307 k) ----------------
// Above the one-launch kernels' size an iteration of three launches costs ~15 us whatever the three kernels move (a dependent
// launch is ~4.5 us end to end on MI355X: hipk_tile_combine_kernel, one wavefront per chunk, averages 4.6 us in the profiles).
// The direction step is folded into the SpMV: K1(k) folds <r,r> of iteration k-1 (every workgroup, the same bits), does the stop
// test and the gamma bookkeeping, forms p_k = r + beta p_{k-1} ON THE FLY at the gathered columns (the owner's formula on the
// owner's operands: the bits the direction kernel would have stored), writes its own rows of p_k into the OTHER p buffer and
// Ap_k, and leaves the per-wavefront tile sums of <p_k, Ap_k>; K2(k) folds them, alpha, x += alpha p_k, r -= alpha Ap_k, <r,r>
// partials.  Same operations, same order per element as the three-launch sequence: same bits
// (tests/test_gpu_api.py::test_cg_two_launch_iteration_is_bit_identical).  General CSR tiles (the FAST form of hipk_spmv_kernel:
// every tile fits the LDS product buffer, no long rows), also for matrices that have a coded form: at these sizes the matrix
// comes from L2 / the Infinity Cache and the kernel is launch-bound either way.
struct hipk_cg2_args {
    const int *crow;
    const int *col;
    const void *val;
    int64_t n;
    int ch, g;
    hipk_cg_scal *scal;
    int64_t it, maxiter;
    const double *part_rr;   // chunk partials of <r,r> (K2 of the iteration before; unused at it == 0)
    const void *r;
    const void *p_old;
    void *p_new;
    void *Ap;
    double *tpart;           // per-wavefront tile sums of <p, Ap>, 4 per tile
};

template <typename T, int CAP>
__global__ __launch_bounds__(HIPK_THREADS) void hipk_cg2_spmv_kernel(hipk_cg2_args a) {
    constexpr int NI = CAP / HIPK_THREADS;
    const int ntiles = (int)((a.n + HIPK_TILE - 1) / HIPK_TILE);
    const int tile = hipk_xcd_tile(blockIdx.x, ntiles);
    __shared__ __attribute__((aligned(16))) T prod[CAP];
    __shared__ int crowL[HIPK_TILE + 1];
    __shared__ double sbuf[HIPK_THREADS];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int *__restrict__ crow = a.crow;
    const int *__restrict__ col = a.col;
    const T *__restrict__ val = (const T *)a.val;
    const T *__restrict__ r = (const T *)a.r;
    const T *__restrict__ po = (const T *)a.p_old;
    hipk_cg_scal *scal = a.scal;
    const int64_t it = a.it;
    const int64_t r0 = (int64_t)(tile < 0 ? 0 : tile) * HIPK_TILE;
    const int nr = tile < 0 ? 0 : (int)((a.n - r0 < HIPK_TILE) ? (a.n - r0) : HIPK_TILE);
    int crow_t = 0, crow_e = 0;
    T rrow = (T)0, prow = (T)0;
    if (t < nr) {
        crow_t = crow[r0 + t];
        rrow = r[r0 + t];
        prow = po[r0 + t];
    }
    if (t == 0 && nr > 0) crow_e = crow[r0 + nr];
    if (it >= scal->stop_it) return;
    // <r,r> of the iteration before -> gamma_it, beta, the stop test of THIS pass (TSL:841, 851-853); every workgroup the same bits
    T beta = (T)0;
    if (it > 0) {
        const double gamma = hipk_reduce_parts(a.part_rr, a.g, sbuf);
        const double gamma_prev = scal->gamma[(it - 1) & 1];
        beta = (T)(gamma / gamma_prev);  // TSL:851
        const bool done = (it >= a.maxiter || gamma <= scal->atol2);
        if (blockIdx.x == 0 && t == 0) {
            scal->gamma[it & 1] = gamma;  // TSL:853
            if (done) scal->stop_it = it;
            hipk_signal(scal->host_sig, done ? (HIPK_SIG_STOP | it) : it);
        }
        if (done) return;
    }
    if (tile < 0) return;
    if (t < nr) crowL[t] = crow_t;
    if (t == 0) crowL[nr] = crow_e;
    __syncthreads();
    const int j0 = crowL[0];
    const int cnt = crowL[nr] - j0;
    if (cnt > 0) {
        const int jb = __builtin_amdgcn_readfirstlane(j0);
        const int *__restrict__ colb = col + jb;
        const T *__restrict__ valb = val + jb;
        unsigned jj[NI];
        int cc[NI];
        T vv[NI], rv[NI], pv[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int j = t + i * HIPK_THREADS;
            jj[i] = (unsigned)(j < cnt ? j : 0);
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) cc[i] = colb[jj[i]];
#pragma unroll
        for (int i = 0; i < NI; ++i) vv[i] = valb[jj[i]];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            rv[i] = r[cc[i]];
            pv[i] = po[cc[i]];
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const T m = beta * pv[i];
            const T pj = rv[i] + m;  // TSL:852 at column cc[i]: the bits its owner stores
            prod[t + i * HIPK_THREADS] = vv[i] * pj;
        }
    }
    __syncthreads();
    double d0 = 0.0;
    if (t < nr) {
        const int lo = crowL[t] - j0, len = crowL[t + 1] - j0 - lo;
        T s = (T)0;
        for (int j = 0; j < len; ++j) s = s + prod[lo + j];
        const T m = beta * prow;
        const T pn = rrow + m;  // TSL:852, own row
        ((T *)a.p_new)[r0 + t] = pn;
        ((T *)a.Ap)[r0 + t] = s;
        d0 = (double)pn * (double)s;
    }
    d0 = hipk_wave_sum(d0);
    if (lane == 0) a.tpart[(size_t)tile * 4 + wave] = d0;
}

// K2: <p,Ap> from the tile sums (the combine kernel's chunk fold, then the spec's fold of the chunk partials), alpha,
// x += alpha p, r -= alpha Ap, partials of <r,r>.  grid = chunks (<= 256).
template <typename T>
__global__ __launch_bounds__(HIPK_THREADS) void hipk_cg2_update_kernel(int64_t n, int ch, int g, const hipk_cg_scal *__restrict__ scal,
                                                                      int64_t it, const double *__restrict__ tpart, int ntiles,
                                                                      const T *__restrict__ Ap, const T *__restrict__ p,
                                                                      T *__restrict__ r, T *__restrict__ x,
                                                                      double *__restrict__ part_rr) {
    const int c = blockIdx.x;
    hipk_pre<T, 2, false> pre;
    pre.issue(n, ch, c, {Ap, (const T *)r});
    if (it >= scal->stop_it) return;
    __shared__ double sbuf[HIPK_THREADS];
    __shared__ double cp[256];
    // chunk partial k = hipk_wave_fold over the chunk's <= 8 tiles (ch = 2048), by ONE thread per chunk: with cnt <= 8 that fold is
    // lane l < cnt holding ((0.0 + tp_l) + 0.0) + 0.0, the other lanes 0.0, and the wavefront tree, whose strides 32, 16, 8 add
    // zeros and whose strides 4, 2, 1 are ((v0 + v4) + (v2 + v6)) + ((v1 + v5) + (v3 + v7)).  (A wavefront per chunk, g / 4 folds
    // in series per workgroup, made this kernel 3 us per 32 chunks slower: 29.5 instead of 17.3 us per iteration at n = 250 k.)
    if ((int)threadIdx.x < g) {
        const int k = threadIdx.x, first = k * 8;
        const int cnt = (ntiles - first < 8) ? ntiles - first : 8;
        double v[8];
#pragma unroll
        for (int l = 0; l < 8; ++l) {
            double tl = 0.0;
            if (l < cnt) {
                const double *w4 = tpart + (size_t)(first + l) * 4;
                tl = 0.0 + ((w4[0] + w4[1]) + (w4[2] + w4[3]));
            }
            v[l] = ((tl + 0.0) + 0.0) + 0.0;   // a[0] + a[2], then + a[1] (+ a[3]) of hipk_wave_fold, then the strides 32, 16, 8
        }
        cp[k] = ((v[0] + v[4]) + (v[2] + v[6])) + ((v[1] + v[5]) + (v[3] + v[7]));
    }
    __syncthreads();
    const double pAp = hipk_reduce_parts(cp, g, sbuf);
    const double gamma = scal->gamma[it & 1];
    const T alpha = (T)(gamma / pAp);  // TSL:846
    double acc = 0.0;
    pre.run([&](int64_t i, int nv, T(&v)[2][hipk_vec<T>::VEC]) {
        constexpr int VEC = hipk_vec<T>::VEC;
        T rv[VEC], xv[VEC], pv[VEC];
        hipk_ld<T>(p, i, nv, pv);
        hipk_ld<T>((const T *)x, i, nv, xv);
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            const T m0 = alpha * pv[k];
            xv[k] = xv[k] + m0;  // TSL:847
            const T m1 = alpha * v[0][k];
            rv[k] = v[1][k] - m1;  // TSL:848
            if (k < nv) acc = fma((double)rv[k], (double)rv[k], acc);  // TSL:850
        }
        hipk_st<T>(x, i, nv, xv);
        hipk_st<T>(r, i, nv, rv);
    });
    acc = hipk_block_sum(acc, sbuf);
    if (threadIdx.x == 0) part_rr[c] = acc;
}

// ---- row-partitioned CG with the exchanges folded into the kernels (hipk_fx.h; hipk_dist.hip drives them) -----------------
// The update / direction kernels above with an exchange in front: workgroups 0 .. world-1 publish this rank's partials (and, in
// the direction kernel, the boundary entries of r) into the peers' mailboxes, every workgroup waits for all sources and folds the
// gathered partials straight from its own mailbox.  grid = max(chunks, world): surplus workgroups only publish and wait.
__global__ __launch_bounds__(HIPK_THREADS) HIPK_SGPR80 void hipk_cg_update_fx_kernel(int64_t n, int ch, int g, const hipk_cg_scal *__restrict__ scal,
                                                                         int64_t it, const double *__restrict__ Ap, double *__restrict__ r,
                                                                         double *__restrict__ part_rr, hipk_fx fx) {
    typedef double T;
    const int c = blockIdx.x;
    hipk_pre<T, 2, false> pre;
    pre.issue(n, ch, c, {Ap, (const T *)r});
    if (it >= scal->stop_it) return;   // the same word on every rank (same partials, same fold): all ranks skip together
    __shared__ double sbuf[HIPK_THREADS];
    __shared__ int fx_ok;
    hipk_fx_publish(fx);
    if (c == 0) hipk_fx_collect(fx, g, sbuf, &fx_ok);
    hipk_fx_await(fx);
    const double pAp = hipk_fx_scalar(fx, 0);
    const double gamma = scal->gamma[it & 1];
    const T alpha = (T)(gamma / pAp);  // TSL:846
    double acc = 0.0;
    pre.run([&](int64_t i, int nv, T(&v)[2][hipk_vec<T>::VEC]) {
        constexpr int VEC = hipk_vec<T>::VEC;
        T rv[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            const T m1 = alpha * v[0][k];
            rv[k] = v[1][k] - m1;  // TSL:848
            if (k < nv) acc = fma((double)rv[k], (double)rv[k], acc);  // TSL:850
        }
        hipk_st<T>(r, i, nv, rv);
    });
    acc = hipk_block_sum(acc, sbuf);
    if (threadIdx.x == 0 && (int64_t)c * ch < n) part_rr[c] = acc;
}

// n = n_ext (own rows + ghost tail), n_own = own rows.  The ghost entries of r arrive with the exchange: the workgroups whose chunk
// reaches into the tail copy their part from the mailbox into r BEFORE they request their operands; everybody else requests first.
__global__ __launch_bounds__(HIPK_THREADS) HIPK_SGPR80 __attribute__((amdgpu_waves_per_eu(8, 8))) void hipk_cg_direction_fx_kernel(int64_t n, int64_t n_own, int ch, int g,
                                                                            hipk_cg_scal *__restrict__ scal, int64_t it, int64_t maxiter,
                                                                            double *__restrict__ r, double *__restrict__ p,
                                                                            double *__restrict__ x, hipk_fx fx) {
    typedef double T;
    const int c = blockIdx.x;
    const bool tail = (int64_t)(c + 1) * ch > n_own && (int64_t)c * ch < n;
    __shared__ double sbuf[HIPK_THREADS];
    __shared__ int fx_ok;
    hipk_pre<T, 2, false> pre;
    if (!tail) pre.issue(n, ch, c, {(const T *)r, (const T *)p});
    if (it >= scal->stop_it) return;
    hipk_fx_publish(fx);
    if (c == 0) hipk_fx_collect(fx, g, sbuf, &fx_ok);
    hipk_fx_await(fx);
    if (tail) {
        const double *halo = hipk_fx_halo(fx);   // fine-grained memory: read from memory, after the collector's flag
        const int64_t lo = ((int64_t)c * ch > n_own ? (int64_t)c * ch : n_own), hi = ((int64_t)(c + 1) * ch < n ? (int64_t)(c + 1) * ch : n);
        for (int64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) r[i] = halo[i - n_own];
        __syncthreads();
        pre.issue(n, ch, c, {(const T *)r, (const T *)p});
    }
    const double pAp = hipk_fx_scalar(fx, 0), rr = hipk_fx_scalar(fx, 1);
    const double gamma = scal->gamma[it & 1];
    const T alpha = (T)(gamma / pAp);  // TSL:846, the same bits hipk_cg_update_fx_kernel derived
    const T beta = (T)(rr / gamma);    // TSL:851
    pre.run([&](int64_t i, int nv, T(&v)[2][hipk_vec<T>::VEC]) {
        constexpr int VEC = hipk_vec<T>::VEC;
        T xv[VEC], pv[VEC];
        hipk_ld<T>((const T *)x, i, nv, xv);
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            const T m0 = alpha * v[1][k];
            xv[k] = xv[k] + m0;  // TSL:847 (with the p of this iteration, before it is replaced)
            const T m = beta * v[1][k];
            pv[k] = v[0][k] + m;  // TSL:852
        }
        hipk_st<T>(x, i, nv, xv);
        hipk_st<T>(p, i, nv, pv);
    });
    if (c == 0 && threadIdx.x == 0) {
        scal->gamma[(it + 1) & 1] = rr;  // TSL:853
        const bool done = (it + 1 >= maxiter || rr <= scal->atol2);  // TSL:841 for the NEXT pass
        if (done) scal->stop_it = it + 1;
        hipk_signal(scal->host_sig, done ? (HIPK_SIG_STOP | (it + 1)) : (it + 1));
    }
}

// hipk_dist.hip's entry points to them (fp64; fx->seq / ch / kind / parts / vec filled by the caller per exchange)
int hipk_cg_update_fx(int64_t n_local, int chunk_rows, int g_red, const void *scal_dev, int64_t it, const void *Ap, void *r,
                      double *part_rr_out, const hipk_fx *fx, hipStream_t stream) {
    int grid = (int)((n_local + chunk_rows - 1) / chunk_rows);
    if (grid < fx->world) grid = fx->world;
    hipk_cg_update_fx_kernel<<<grid, HIPK_THREADS, 0, stream>>>(n_local, chunk_rows, g_red, (const hipk_cg_scal *)scal_dev, it,
                                                               (const double *)Ap, (double *)r, part_rr_out, *fx);
    HIPK_CHECK_HIP(hipGetLastError());
    return HIPK_OK;
}
int hipk_cg_direction_fx(int64_t n_ext, int64_t n_local, int chunk_rows, int g_red, void *scal_dev, int64_t it, int64_t maxiter, void *r,
                         void *p, void *x, const hipk_fx *fx, hipStream_t stream) {
    int grid = (int)((n_ext + chunk_rows - 1) / chunk_rows);
    if (grid < fx->world) grid = fx->world;
    hipk_cg_direction_fx_kernel<<<grid, HIPK_THREADS, 0, stream>>>(n_ext, n_local, chunk_rows, g_red, (hipk_cg_scal *)scal_dev, it, maxiter,
                                                                  (double *)r, (double *)p, (double *)x, *fx);
    HIPK_CHECK_HIP(hipGetLastError());
    return HIPK_OK;
}

// Streaming policy on one device (vectors in HBM: N >> 8 M rows): the direction step as TWO launches -- alpha, beta, the next gamma
// and the stop test once, by one workgroup (the same fold of the same partials: the same bits); then a FLAT grid of short
// workgroups (2048 elements each, every load requested before the stop word is read), which a step without a dot is free to use.
// Why: at N = 64 M a workgroup per 256 KB chunk keeps 1954 x 5 distant streams open; the same bytes move 6-10 % faster from short
// workgroups over adjacent addresses, and lose less when the vectors landed badly (profiles/r02_axpy_probe_64m.txt,
// r02_axpy_realloc_64m.txt: 5.8 -> 6.1 TB/s, and 4.9 -> 5.4 TB/s in the slow placement); per workgroup the fold of 2 x 1954
// partials is what stood in the way.  The extra launch costs ~4 us of an iteration of ~1 ms.
__global__ __launch_bounds__(HIPK_THREADS) void hipk_cg_scalars_kernel(int g, hipk_cg_scal *__restrict__ scal, int64_t it, int64_t maxiter,
                                                                       const double *__restrict__ part_pAp,
                                                                       const double *__restrict__ part_rr) {
    if (it >= scal->stop_it) return;
    __shared__ double sbuf[2 * HIPK_THREADS];
    double pAp, rr;
    hipk_reduce_parts2(part_pAp, part_rr, g, pAp, rr, sbuf);
    if (threadIdx.x == 0) {
        const double gamma = scal->gamma[it & 1];
        scal->dir_alpha = gamma / pAp;       // TSL:846, the same bits hipk_cg_update_kernel derived
        scal->dir_beta = rr / gamma;         // TSL:851
        scal->gamma[(it + 1) & 1] = rr;      // TSL:853
        const bool done = (it + 1 >= maxiter || rr <= scal->atol2);  // TSL:841 for the NEXT pass (this pass compares against `it`)
        if (done) scal->stop_it = it + 1;
        hipk_signal(scal->host_sig, done ? (HIPK_SIG_STOP | (it + 1)) : (it + 1));
    }
}

template <typename T>
__global__ __launch_bounds__(HIPK_THREADS) void hipk_cg_direction_flat_kernel(int64_t n, const hipk_cg_scal *__restrict__ scal, int64_t it,
                                                                              const T *__restrict__ r, T *__restrict__ p,
                                                                              T *__restrict__ x) {
    constexpr int VEC = hipk_vec<T>::VEC;
    constexpr int STEPS = HIPK_BASE_CHUNK / (VEC * HIPK_THREADS);
    const int64_t base = (int64_t)blockIdx.x * HIPK_BASE_CHUNK + (int64_t)VEC * threadIdx.x;
    T rv[STEPS][VEC], pv[STEPS][VEC], xv[STEPS][VEC];
    int nvs[STEPS];
#pragma unroll
    for (int k = 0; k < STEPS; ++k) {
        const int64_t i = base + (int64_t)k * VEC * HIPK_THREADS;
        nvs[k] = (i < n) ? ((n - i < VEC) ? (int)(n - i) : VEC) : 0;
        if (nvs[k] > 0) {
            hipk_ld_nt_vec<T>(r, i, nvs[k], rv[k]);
            hipk_ld_nt_vec<T>((const T *)p, i, nvs[k], pv[k]);
            hipk_ld_nt_vec<T>((const T *)x, i, nvs[k], xv[k]);
        }
    }
    if (it >= scal->stop_it) return;  // hipk_cg_scalars_kernel of THIS pass has run: it sets stop_it = it + 1 at the earliest
    const T alpha = (T)scal->dir_alpha;
    const T beta = (T)scal->dir_beta;
#pragma unroll
    for (int k = 0; k < STEPS; ++k) {
        if (nvs[k] > 0) {
            const int64_t i = base + (int64_t)k * VEC * HIPK_THREADS;
            T xo[VEC], po[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const T m0 = alpha * pv[k][e];
                xo[e] = xv[k][e] + m0;  // TSL:847 (with the p of this iteration, before it is replaced)
                const T m = beta * pv[k][e];
                po[e] = rv[k][e] + m;  // TSL:852
            }
            hipk_st_nt_vec<T>(x, i, nvs[k], xo);  // x is not read again before the next direction step
            hipk_st<T>(p, i, nvs[k], po);
        }
    }
}

// ---- placement probe (include/hipk.h: hipk_placement_probe).  At N = 64 M the direction step above runs at one of two discrete
// speeds -- 5.85 or 4.9 TB/s -- depending on where the three vectors landed PHYSICALLY (profiles/r02_axpy_realloc_64m.txt: the same
// kernel at the same virtual addresses, re-allocated; vectors in separate allocations were slow every time, vectors in ONE
// allocation fast in about half of the draws).  This kernel has the step's memory shape (reads r, p, x non-temporal, writes p and
// x) and stores back the bits it loaded, so it can run on live vectors; the host times it and re-draws the allocation when it
// reads the slow level.
__device__ __forceinline__ bool hipk_value_bits_eq(double a, double b) { return __double_as_longlong(a) == __double_as_longlong(b); }
__device__ __forceinline__ bool hipk_value_bits_eq(float a, float b) { return __float_as_int(a) == __float_as_int(b); }
template <typename T>
__global__ __launch_bounds__(HIPK_THREADS) void hipk_probe3_kernel(int64_t n, const T *__restrict__ r, T *__restrict__ p, T *__restrict__ x,
                                                                   T never) {
    constexpr int VEC = hipk_vec<T>::VEC;
    constexpr int STEPS = HIPK_BASE_CHUNK / (VEC * HIPK_THREADS);
    const int64_t base = (int64_t)blockIdx.x * HIPK_BASE_CHUNK + (int64_t)VEC * threadIdx.x;
    T rv[STEPS][VEC], pv[STEPS][VEC], xv[STEPS][VEC];
    int nvs[STEPS];
#pragma unroll
    for (int k = 0; k < STEPS; ++k) {
        const int64_t i = base + (int64_t)k * VEC * HIPK_THREADS;
        nvs[k] = (i < n) ? ((n - i < VEC) ? (int)(n - i) : VEC) : 0;
        if (nvs[k] > 0) {
            hipk_ld_nt_vec<T>(r, i, nvs[k], rv[k]);
            hipk_ld_nt_vec<T>((const T *)p, i, nvs[k], pv[k]);
            hipk_ld_nt_vec<T>((const T *)x, i, nvs[k], xv[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < STEPS; ++k) {
        if (nvs[k] > 0) {
            const int64_t i = base + (int64_t)k * VEC * HIPK_THREADS;
            // `never` is a NaN with a payload no computation produces: the comparison keeps the loads of r alive, the stores put
            // back what was loaded
#pragma unroll
            for (int e = 0; e < VEC; ++e)
                if (hipk_value_bits_eq(rv[k][e], never)) pv[k][e] = rv[k][e];
            hipk_st_nt_vec<T>(x, i, nvs[k], xv[k]);
            hipk_st<T>(p, i, nvs[k], pv[k]);
        }
    }
}

extern "C" int hipk_placement_probe(int64_t n, const void *r, void *p, void *x, int dtype, int reps, double *us_out,
                                    hipk_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    HIPK_REQUIRE(n > 0 && r && p && x && us_out, HIPK_ERR_ARG, "null argument");
    HIPK_REQUIRE(dtype == HIPK_F64 || dtype == HIPK_F32, HIPK_ERR_UNSUPPORTED, "dtype must be f32/f64");
    HIPK_REQUIRE(hipk_aligned16(r) && hipk_aligned16(p) && hipk_aligned16(x), HIPK_ERR_ALIGN, "vectors must be 16-byte aligned");
    if (reps < 1) reps = 1;
    if (reps > 16) reps = 16;
    const unsigned grid = (unsigned)((n + HIPK_BASE_CHUNK - 1) / HIPK_BASE_CHUNK);
    hipk_event_pair ev[16];
    for (int i = 0; i < reps; ++i) HIPK_CHECK_HIP(ev[i].create());
    // one untimed pass (first touch, TLB), then `reps` passes each with events bound to its dispatch (hipk_solve.h)
    for (int i = -1; i < reps; ++i) {
        void *argv[5];
        double nan64;
        float nan32;
        const unsigned long long b64 = 0x7FF8DEADBEEF1234ull;
        const unsigned b32 = 0x7FC0BEEFu;
        memcpy(&nan64, &b64, 8);
        memcpy(&nan32, &b32, 4);
        argv[0] = (void *)&n;
        argv[1] = (void *)&r;
        argv[2] = (void *)&p;
        argv[3] = (void *)&x;
        argv[4] = dtype == HIPK_F64 ? (void *)&nan64 : (void *)&nan32;
        const void *fn = dtype == HIPK_F64 ? (const void *)hipk_probe3_kernel<double> : (const void *)hipk_probe3_kernel<float>;
        if (i < 0)
            HIPK_CHECK_HIP(hipLaunchKernel(fn, dim3(grid), dim3(HIPK_THREADS), argv, 0, stream));
        else
            HIPK_CHECK_HIP(hipExtLaunchKernel(fn, dim3(grid), dim3(HIPK_THREADS), argv, 0, stream, ev[i].a, ev[i].b, 0));
    }
    HIPK_CHECK_HIP(hipStreamSynchronize(stream));
    double best = 1e300;
    for (int i = 0; i < reps; ++i) {
        float ms = 0.f;
        HIPK_CHECK_HIP(hipEventElapsedTime(&ms, ev[i].a, ev[i].b));
        if ((double)ms * 1e3 < best) best = (double)ms * 1e3;
    }
    *us_out = best;
    return HIPK_OK;
}

// x += alpha p alone (TSL:847): the row-partitioned solver runs it on a side stream while the <r,r> / halo collective of the
// iteration is in flight; alpha from the same partials and the same gamma as the update and direction kernels (same bits)
template <typename T>
__global__ __launch_bounds__(HIPK_THREADS) void hipk_cg_xupdate_kernel(int64_t n, int ch, int g, const hipk_cg_scal *__restrict__ scal,
                                                                       int64_t it, const double *__restrict__ part_pAp,
                                                                       const T *__restrict__ p, T *__restrict__ x) {
    const int c = blockIdx.x;
    hipk_pre<T, 2> pre;
    pre.issue(n, ch, c, {p, (const T *)x});
    if (it >= scal->stop_it) return;
    __shared__ double sbuf[HIPK_THREADS];
    const double pAp = hipk_reduce_parts(part_pAp, g, sbuf);
    const T alpha = (T)(scal->gamma[it & 1] / pAp);  // TSL:846
    pre.run([&](int64_t i, int nv, T(&v)[2][hipk_vec<T>::VEC]) {
        constexpr int VEC = hipk_vec<T>::VEC;
        T xv[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            const T m0 = alpha * v[0][k];
            xv[k] = v[1][k] + m0;
        }
        hipk_st<T>(x, i, nv, xv);
    });
}

// =====================================================================================================================
// Small systems (<= 8 reduction chunks = n <= 16384, rows of <= 12 entries): THE WHOLE CG LOOP IN ONE LAUNCH.
// Three launches of >= 4.9 us per iteration for 80 KB vectors leave such systems launch-bound (16 us per iteration at n = 10^4).
// Here 8 g workgroups stay resident on one XCD and meet at two hand-offs per iteration (csrc/hipk_handoff.h; the scheme of
// hipk_gm_solve_lds_kernel).  Every thread plays two roles:
//   * VECTOR role: row (u, e8) of the spec's virtual-thread layout (sub-workgroup s of chunk c owns the virtual threads s + 8u):
//     x, r, p of that row live in REGISTERS for the whole solve; <r,r> is the chain of its virtual thread + a 32-lane tree, published
//     as a sub-partial that every consumer folds with the last three levels of the chunk tree -- the bits of the chunk dot;
//   * SpMV role: row tile*256 + tid of ONE 256-row tile, because <p,Ap> is the TILED dot of the SpMV epilogue (wavefront sums over
//     64 contiguous rows).  The row's matrix entries and the p values they multiply stay in registers: after the <r,r> hand-off
//     the thread gathers r at its columns and advances its own copies, p_j = r_j + beta p_j -- the owner's formula on the owner's
//     operands, the same bits -- so p itself is never exchanged.
// hand-off 1: tile sums of <p,Ap> + Ap by tile rows;  hand-off 2: sub-partials of <r,r> + r.  Arithmetic per element = the
// three-kernel loop's (TSL:845-853), bit for bit.
template <typename T>
struct hipk_cg_lds_args {
    int64_t n;
    int g;
    const int *crow;
    const int *col;
    const T *val;
    T *x, *r, *p;
    T *Ap;               // exchange buffer: A p by rows
    hipk_lds_ctl *ctl;          // progress of the launch
    double *gamma;              // [2] by iteration parity; in: gamma of iteration it0 (it0 > 0 or M = identity), out: of the last one
    const double *atol2;        // device scalars of the solve's header block
    int64_t *stop_it;
    const T *dinv;              // PRE: the Jacobi preconditioner's diagonal (M = diag(dinv), TSL:849)
    const double *rz0_parts;    // PRE, it0 = 0: chunk partials of gamma0 = <r0, M r0> (hipk_pcg_start_kernel)
    double *tile_pp;     // [ntiles * 4] wavefront sums of <p,Ap>
    double *rr_sub;      // [8 g] sub-partials of <r,r>
    double *rz_sub;      // PRE: [8 g] sub-partials of <r, M r>
    unsigned long long *flag_a, *flag_b;   // [64] each, zeroed before the launch
    int64_t it0;         // iterations done before this launch
    int64_t maxiter;
    int64_t max_its;     // iteration budget of one launch
    int test_not_resident;   // tests (HIPK_TEST_LDS_NOT_RESIDENT): report the placement check as failed
    int spread;              // more than 64 workgroups: one per block all over the chip (then LOCAL = false)
};
static constexpr int kCgRowRegs = 12;

template <typename T, bool LOCAL, bool PRE>
__global__ __launch_bounds__(HIPK_THREADS, 2) void hipk_cg_solve_lds_kernel(hipk_cg_lds_args<T> a) {
    constexpr int VEC = hipk_vec<T>::VEC;
    int wg = blockIdx.x;                             // spread (more than 64 workgroups): one per block, anywhere on the chip
    if (!a.spread) {
        if (blockIdx.x & 7) return;                  // the working blocks share an XCD (dispatch is round-robin over 8)
        wg = blockIdx.x >> 3;
    }
    const int c = wg / kGmSub, s = wg % kGmSub;
    const int g = a.g, nwg = g * kGmSub;
    if (c >= g) return;
    hipk_lds_ctl *scal = a.ctl;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int u = tid & 31, e8 = tid >> 5;
    const int64_t n = a.n;
    const int64_t base = (int64_t)c * HIPK_BASE_CHUNK;
    const int64_t row = base + (int64_t)VEC * (s + kGmSub * u) + (int64_t)(e8 / VEC) * (VEC * HIPK_THREADS) + (e8 % VEC);
    const bool live = row < n;
    const int ntiles = (int)((n + HIPK_TILE - 1) / HIPK_TILE);
    const int tile = c * (HIPK_BASE_CHUNK / HIPK_TILE) + s;
    const int64_t trow = (int64_t)tile * HIPK_TILE + tid;
    const bool tlive = trow < n;

    __shared__ T wl[HIPK_THREADS], wz[HIPK_THREADS];
    __shared__ double bc[4];
    __shared__ int fail;
    __shared__ unsigned long long res_lds;
    if (tid == 0) fail = 0;

    // vector role
    T x_own = live ? a.x[row] : (T)0, r_own = live ? a.r[row] : (T)0, p_own = live ? a.p[row] : (T)0;
    // SpMV role: the tile row's entries, and p at its columns (the launches before this one left p in memory)
    int lo = 0, len = 0;
    if (tlive) {
        lo = a.crow[trow];
        len = a.crow[trow + 1] - lo;
    }
    unsigned cj[kCgRowRegs];
    T vj[kCgRowRegs], pg[kCgRowRegs];
#pragma unroll
    for (int j = 0; j < kCgRowRegs; ++j) {
        const int cc = (j < len) ? a.col[lo + j] : 0;
        cj[j] = (unsigned)cc * (unsigned)sizeof(T);
        vj[j] = (j < len) ? a.val[lo + j] : (T)0;
        pg[j] = (j < len) ? a.p[cc] : (T)0;
    }
    T p_t = tlive ? a.p[trow] : (T)0;
    // PRE: the diagonal of M at the own row, at the tile row and at the tile row's columns
    T d_own = (T)1, d_t = (T)1, dc[kCgRowRegs];
#pragma unroll
    for (int j = 0; j < kCgRowRegs; ++j) dc[j] = (T)1;
    if (PRE) {
        d_own = live ? a.dinv[row] : (T)0;
        d_t = tlive ? a.dinv[trow] : (T)0;
#pragma unroll
        for (int j = 0; j < kCgRowRegs; ++j) dc[j] = (j < len) ? a.dinv[a.col[lo + j]] : (T)0;
    }
    int wmax = len < kCgRowRegs ? len : kCgRowRegs;
    for (int off = 32; off > 0; off >>= 1) {
        const int o = __shfl_xor(wmax, off);
        wmax = o > wmax ? o : wmax;
    }
    wmax = __builtin_amdgcn_readfirstlane(wmax);
    double gamma = a.gamma[a.it0 & 1];
    const double atol2 = *a.atol2;
    const int64_t stop0 = *a.stop_it;
    double rs_last = scal->rs_last;

    // every workgroup resident (and, LOCAL, on one XCD)?  Nothing has been modified yet: a failure leaves the solve to the launches
    int epoch = 0;
    if (LOCAL && tid == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        __hip_atomic_fetch_or(&scal->xcc_mask, 1u << (xcc & 15u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!hipk_gbar(&scal->bar, nwg, epoch, &fail) || a.test_not_resident) {
        if (tid == 0) scal->redo = -1;
        return;
    }
    if (LOCAL) {
        if (tid == 0) {
            const unsigned mask = __hip_atomic_load(&scal->xcc_mask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            fail = (__builtin_popcount(mask) == 1) ? 0 : 1;
        }
        __syncthreads();
        if (fail) {
            if (tid == 0) scal->redo = -2;
            return;
        }
    }
    if (PRE && a.it0 == 0) {   // gamma0 = <r0, M r0>: chunk partials of the launch before this one (hipk_reduce_parts)
        if (tid < 64) {
            double a8 = (tid < g) ? a.rz0_parts[tid] : 0.0;
            a8 = 0.0 + a8;
            a8 = hipk_wave_sum(a8);
            if (tid == 0) bc[2] = a8;
        }
        __syncthreads();
        gamma = bc[2];
    }
    unsigned long long seq = 0;
    int64_t it = a.it0;
    bool done = stop0 <= it;
    while (!done) {
        // ---- SpMV role: (A p) of the tile row, wavefront sums of p .* (A p)  (TSL:845-846)
        T acc_row = (T)0;
#pragma unroll
        for (int j = 0; j < kCgRowRegs; ++j)
            if (j < wmax) {
                const T pr = vj[j] * pg[j];
                acc_row = (j < len) ? acc_row + pr : acc_row;
            }
        const T Ap_t = tlive ? acc_row : (T)0;
        double d0 = tlive ? (double)p_t * (double)Ap_t : 0.0;
        d0 = hipk_wave_sum(d0);
        if (lane == 0 && tile < ntiles) hipk_ho_store<LOCAL>(&a.tile_pp[(size_t)tile * 4 + wave], d0);
        if (tlive) hipk_ho_store<LOCAL>(a.Ap + trow, Ap_t);
        if (hipk_ho_sync<LOCAL>(a.flag_a, wg, nwg, ++seq, 0u, &res_lds) == ~0ull) {
            if (tid == 0) scal->redo = -3;   // cannot happen once every workgroup has passed the placement check
            return;
        }
        // ---- vector role: alpha, r, x, <r,r> sub-partial  (TSL:846-850)
        const T Ap_own = live ? hipk_peek_t<T>(a.Ap + row) : (T)0;
        if (tid < 64) {
            const double *tp = a.tile_pp;
            const double pAp = hipk_fold_64x8(tid, g, [&](int ci, int tt) {
                const int tl = ci * (HIPK_BASE_CHUNK / HIPK_TILE) + tt;
                if (tl >= ntiles) return 0.0;
                const double *w4 = tp + (size_t)tl * 4;
                const double w0 = hipk_peek(w4), w1 = hipk_peek(w4 + 1), w2 = hipk_peek(w4 + 2), w3 = hipk_peek(w4 + 3);
                return 0.0 + ((w0 + w1) + (w2 + w3));
            });
            if (tid == 0) bc[0] = pAp;
        }
        __syncthreads();
        const T alpha = (T)(gamma / bc[0]);
        {
            const T m1 = alpha * Ap_own;
            r_own = r_own - m1;
            const T m0 = alpha * p_own;
            x_own = x_own + m0;
        }
        wl[tid] = r_own;
        if (PRE) wz[tid] = d_own * r_own;   // z = M r (TSL:849), never stored to memory
        if (live) hipk_ho_store<LOCAL>(a.r + row, r_own);
        __syncthreads();
        if (tid < 32) {
            double acc = 0.0, acc1 = 0.0;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const double v = (double)wl[e * 32 + tid];
                acc = fma(v, v, acc);
                if (PRE) acc1 = fma(v, (double)wz[e * 32 + tid], acc1);   // TSL:850
            }
            acc = hipk_half_sum(acc);
            if (PRE) acc1 = hipk_half_sum(acc1);
            if (tid == 0) {
                hipk_ho_store<LOCAL>(&a.rr_sub[wg], acc);
                if (PRE) hipk_ho_store<LOCAL>(&a.rz_sub[wg], acc1);
            }
        }
        if (hipk_ho_sync<LOCAL>(a.flag_b, wg, nwg, ++seq, 0u, &res_lds) == ~0ull) {
            if (tid == 0) scal->redo = -3;
            return;
        }
        // ---- beta, p (both roles), stop test  (TSL:851-853, 841)
        T rg[kCgRowRegs];
#pragma unroll
        for (int j = 0; j < kCgRowRegs; ++j)
            if (j < wmax) rg[j] = hipk_peek_off<T>(a.r, cj[j]);
        const T r_t = tlive ? hipk_peek_t<T>(a.r + trow) : (T)0;
        if (tid < 64) {
            const double rr = hipk_fold_64x8(tid, g, [&](int ci, int ss) { return hipk_peek(a.rr_sub + ci * kGmSub + ss); });
            if (tid == 0) bc[1] = rr;
        } else if (PRE && tid < 128) {
            const double rz = hipk_fold_64x8(tid - 64, g, [&](int ci, int ss) { return hipk_peek(a.rz_sub + ci * kGmSub + ss); });
            if (tid == 64) bc[3] = rz;
        }
        __syncthreads();
        const double rr = bc[1];
        const double gamma_new = PRE ? bc[3] : rr;   // <r,z> steers alpha and beta, <r,r> the stop test (TSL:835-841)
        const T beta = (T)(gamma_new / gamma);
        {
            const T z_own = PRE ? d_own * r_own : r_own;
            const T m = beta * p_own;
            p_own = z_own + m;
            const T z_t = PRE ? d_t * r_t : r_t;
            const T mt = beta * p_t;
            p_t = z_t + mt;
        }
#pragma unroll
        for (int j = 0; j < kCgRowRegs; ++j)
            if (j < wmax) {
                const T zj = PRE ? dc[j] * rg[j] : rg[j];
                const T m = beta * pg[j];
                pg[j] = zj + m;
            }
        gamma = gamma_new;
        rs_last = rr;
        ++it;
        done = (it >= a.maxiter || rr <= atol2);
        if (it - a.it0 >= a.max_its) break;
    }
    if (live) {
        a.x[row] = x_own;
        a.p[row] = p_own;
    }
    if (wg == 0 && tid == 0) {
        a.gamma[it & 1] = gamma;
        scal->rs_last = rs_last;
        scal->it_done = it;
        if (done && it < stop0) *a.stop_it = it;
    }
}

// res2 = sum parts0, xx = sum parts1 -> scal
__global__ __launch_bounds__(HIPK_THREADS) void hipk_cg_final_kernel(hipk_cg_scal *__restrict__ scal, int g,
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


// two launches per iteration (hipk_cg2_*): 33 .. 150 reduction chunks.  Same box, alternating, 5-point Poisson, us per iteration
// two / three launches: n = 90 k 12.4 / 15.7-16.7, 160 k 14.0 / 16.7-17.5, 250 k 15.7 / 17.0-17.9, 360 k 18.5 / 18.7-18.8,
// 518 k 21.2 / 20.2 (the general CSR tiles' 12 bytes per entry catch up with the saved launch): profiles/r03_cg_two_launch.txt
static constexpr int kCg2MaxChunks = 150;

extern "C" size_t hipk_cg_work_bytes(int64_t n, int dtype) {
    const size_t sv = (dtype == HIPK_F64) ? 8 : 4;
    const size_t vec = hipk_align_up((size_t)(n > 0 ? n : 1) * sv, 256);
    // r, p, Ap; mid-size systems (two launches per iteration) a second p: the direction step is formed while the old p is gathered
    const hipk_geom gm = hipk_make_geom(n > 0 ? n : 1);
    // (the one-launch mid-size loop, hipk_cg_mid.h, keeps r as 16-byte flagged words in Ap + that fourth vector)
    const bool mid = gm.g > kMidMinChunks && gm.g <= kMidMaxChunks;   // + the chunk-partial slots of that loop, a line each
    // (r travels as 16-byte flagged words whatever the dtype: 16 n bytes from Ap on -- one more fp64 vector, three more fp32 ones)
    const size_t ll = hipk_align_up((size_t)(n > 0 ? n : 1) * 16, 256);
    return 256 + hipk_scratch_bytes() + 3 * vec + (mid ? (ll - vec) + kMidSlotBytes : 0);
}

template <typename T>
static int hipk_cg_solve_t(hipk_csr_s *A, const T *b, T *x, char *work, const hipk_params *prm, hipk_stats *st,
                           hipStream_t stream) {
    const int64_t n = A->n_rows;
    const hipk_geom gm = A->geom;
    const size_t vec = hipk_align_up((size_t)(n > 0 ? n : 1) * sizeof(T), 256);
    hipk_cg_scal *scal = (hipk_cg_scal *)work;
    double *parts = (double *)(work + 256);
    double *part_a = parts;                       // <p,Ap> / <b,b> / <x,x>
    double *part_b = parts + HIPK_MAX_PARTS;      // <r,r>
    double *part_c = parts + 2 * HIPK_MAX_PARTS;  // spare dot slot of the spmv kernel
    T *r = (T *)(work + 256 + hipk_scratch_bytes());
    T *p = (T *)((char *)r + vec);
    T *Ap = (T *)((char *)p + vec);

    const int64_t maxiter = (prm->maxiter < 0) ? 10 * n : prm->maxiter;  // TSL:982-984
    // torch.square(torch.tensor(tol)): python floats become fp32 tensors (TSL:816-817)
    const float tolf = (float)prm->tol, atolf = (float)prm->atol;
    const double tol2 = (double)(tolf * tolf), atol_sq = (double)(atolf * atolf);
    int64_t check = prm->check_every > 0 ? prm->check_every : 64;

    hipk_event_pair whole;
    HIPK_CHECK_HIP(whole.create());
    hipk_spmv_profiler prof(prm->profile);   // only launches of the selected kind carry events (hipk_solve.h: chain mode perturbs)
    HIPK_CHECK_HIP(hipEventRecord(whole.a, stream));

    hipk_spmv_args sa;
    memset(&sa, 0, sizeof(sa));
    sa.crow = A->crow;
    sa.col = A->col;
    sa.val = A->val;
    sa.n = n;
    sa.ch = gm.ch;
    sa.g = gm.g;
    int rc;
    int64_t matvecs = 0;

    // r0 = b - A x0 with <r0,r0> partials (TSL:820, 826); <b,b> partials (TSL:815)
    sa.x = x;
    sa.y = r;
    sa.mode = HIPK_SPMV_RESID | HIPK_SPMV_DOT_YY;
    sa.bsub = b;
    sa.part0 = part_c;
    sa.part1 = part_b;
    if ((rc = hipk_launch_spmv(A, sa, stream)) != HIPK_OK) return rc;
    ++matvecs;
    if ((rc = hipk_launch_dot_parts(n, b, b, A->dtype, part_a, stream)) != HIPK_OK) return rc;
    hipk_pacer pace(A->host_poll, &scal->stop_it, check);
    HIPK_CHECK_HIP(pace.create());
    hipk_cg_start_kernel<T><<<gm.g, HIPK_THREADS, 0, stream>>>(n, gm.ch, gm.g, scal, part_b, part_a, r, p, tol2,
                                                                atol_sq, maxiter, pace.device_sig());
    HIPK_CHECK_HIP(hipGetLastError());

    // ---- iteration loop: the host enqueues iterations a few ahead of the GPU and stops when the direction kernel
    // reports the stop (hipk_pacer, hipk_solve.h); launches past the stop are no-ops on the device.
    sa.x = p;
    sa.y = Ap;
    sa.mode = HIPK_SPMV_DOT_W;
    sa.w = p;
    sa.bsub = nullptr;
    sa.part0 = part_a;
    sa.part1 = part_c;
    sa.stop_it = &scal->stop_it;
    // launch-bound systems (<= 8 reduction chunks): no combine launch, the vector kernels fold the SpMV's tile sums
    const bool small = gm.g <= 8 && !getenv("HIPK_CG_NO_SMALL");
    const int ntiles = (int)((n + HIPK_TILE - 1) / HIPK_TILE);
    sa.skip_combine = small ? 1 : 0;
    // x, r, p, Ap beyond 1.5 x the 256 MiB Infinity Cache: the vector kernels treat every operand as a stream
    // (HIPK_CG_STREAMS=0/1 forces the choice: A/B measurements)
    bool streams = 4 * (size_t)n * sizeof(T) > (size_t)384 << 20;
    if (const char *e = getenv("HIPK_CG_STREAMS")) streams = e[0] == '1';
    // with it, and a vector alone beyond the 256 MiB Infinity Cache, the direction step as a scalars launch + a flat grid
    // (same vectors, same process, per CG iteration: N = 64 M 1079 -> 1011 us; N = 32 M 473 -> 471; N = 16 M, where p still finds
    // room in that cache, 237 -> 246: not taken there).  HIPK_CG_FLAT_DIRECTION=0|1 forces (tools/flat_probe.py, tests)
    bool flat_dir = (size_t)n * sizeof(T) > ((size_t)256 << 20);
    if (const char *e = getenv("HIPK_CG_FLAT_DIRECTION")) flat_dir = e[0] == '1';

    int64_t it = 0, stop = INT64_MAX;
    // launch-bound systems of 9 .. 512 chunks (fp64, rows of <= 12 entries within a window around their chunk): the whole loop in
    // one launch, one workgroup per chunk or pair of chunks (hipk_cg_mid.h); HIPK_CG_MID=0 leaves them to the paths below.
    // (At 9 .. 32 chunks it replaces the eight-workgroups-per-chunk kernel below: 5.0 against 10.7 us per iteration at n = 40 000.)
    static bool mid_failed = false;
    bool mid_loop = false;
    {
        mid_loop = it == 0 && gm.g > kMidMinChunks && gm.g <= kMidMaxChunks && gm.ch == HIPK_BASE_CHUNK && A->op_cb == nullptr &&
                   A->crow != nullptr && A->max_row_len <= 12 && prm->profile == 0 && maxiter > 0 && !mid_failed &&
                   !(getenv("HIPK_CG_MID") && getenv("HIPK_CG_MID")[0] == '0') && !getenv("HIPK_CG_NO_LDS_LOOP") && !getenv("HIPK_CG_NO_SMALL");
        // one workgroup of 1024 threads per CU; a chunk each up to n_cu chunks, two each beyond
        const int nch = gm.g <= A->n_cu ? 1 : 2;
        void (*mid_kern)(hipk_cg_mid_args) =
            nch == 1 ? (A->max_row_len <= 5   ? hipk_cg_mid_kernel<T, 5, 1>
                        : A->max_row_len <= 7 ? hipk_cg_mid_kernel<T, 7, 1>
                        : A->max_row_len <= 9 ? hipk_cg_mid_kernel<T, 9, 1>
                                              : hipk_cg_mid_kernel<T, 12, 1>)
                     : (A->max_row_len <= 5 ? hipk_cg_mid_kernel<T, 5, 2> : hipk_cg_mid_kernel<T, 7, 2>);
        const int mid_threads = 1024, mid_grid = (gm.g + nch - 1) / nch;
        if (nch == 2 && A->max_row_len > 7) mid_loop = false;   // four rows per thread: at most 7 entries each in registers
        size_t lds = 0;
        hipk_mid_plan plan;
        memset(&plan, 0, sizeof(plan));
        if (mid_loop) {
            // once per handle: the 256-column tiles each workgroup's window holds (hipk_mid.h); not for matrices whose rows reach
            // further than the plan's range, or whose windows do not fit the LDS
            mid_loop = hipk_mid_plan_get(A, nch, stream, &plan);
            lds = mid_loop ? hipk_cg_mid_lds_bytes(plan.max_slots * HIPK_TILE, nch, false, sizeof(T)) : 0;
            int occ = 0;
            mid_loop = mid_loop && plan.max_slots <= kMidPlanSlots && lds <= (size_t)160 * 1024 &&
                       hipFuncSetAttribute((const void *)mid_kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess &&
                       hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, mid_kern, mid_threads, lds) == hipSuccess &&
                       (int64_t)occ * A->n_cu >= mid_grid;
            (void)hipGetLastError();
        }
        if (mid_loop) {
            const size_t vec8 = hipk_align_up((size_t)n * 16, 256) / 2;   // half of the flagged words' 16 n bytes
            const char *e = getenv("HIPK_CG_LAUNCH_ITS");
            hipk_cg_mid_args ca;
            memset(&ca, 0, sizeof(ca));
            ca.n = n;
            ca.g = gm.g;
            ca.win = plan.max_slots * HIPK_TILE;
            ca.plan = plan;
            ca.crow = A->crow;
            ca.col = A->col;
            ca.val = A->val;
            ca.x = x;
            ca.r = r;
            ca.p = p;
            ca.r_ll = (unsigned long long *)Ap;      // Ap + the fourth vector: 2 x vec >= 16 n bytes
            ca.pap_ll = (unsigned long long *)((char *)Ap + 2 * vec8);   // behind the four vectors (hipk_cg_work_bytes)
            ca.rr_ll = ca.pap_ll + kMidSlotBytes / 16;
            // a 256-byte line per chunk partial: every workgroup polls every slot, and packed slots are ONE memory channel's
            // hot spot (same box, us per iteration, 256 B / 16 B per slot: 5.1 / 5.8 at 79 chunks, 5.4 / 6.6 at 123, 6.8 / 8.2 at
            // 254; 64 B from 257 chunks: 9.65 / 9.95 at 489; 16 B up to 32 chunks: 5.05 / 5.3 at 20) -- HIPK_CG_MID_STRIDE forces
            ca.slot_stride = getenv("HIPK_CG_MID_STRIDE") ? atoi(getenv("HIPK_CG_MID_STRIDE")) : gm.g <= 32 ? 1 : gm.g <= 256 ? 16 : 4;
            if (ca.slot_stride < 1 || ca.slot_stride > 16) ca.slot_stride = 16;
            ca.ctl = &scal->ctl;
            ca.gamma = scal->gamma;
            ca.atol2 = &scal->atol2;
            ca.stop_it = &scal->stop_it;
            ca.maxiter = maxiter;
            ca.xcd_aware = !(getenv("HIPK_CG_MID_XCD") && getenv("HIPK_CG_MID_XCD")[0] == '0');
            ca.max_its = e ? atoll(e) : 16384;
            if (ca.max_its < 1) ca.max_its = 1;
            const int fail_launch = getenv("HIPK_TEST_LDS_NOT_RESIDENT") ? (atoi(getenv("HIPK_TEST_LDS_NOT_RESIDENT")) > 1 ? atoi(getenv("HIPK_TEST_LDS_NOT_RESIDENT")) : 1) : 0;
            int launch_no = 0;
            hipk_cg_scal hs0;
            for (;;) {
                ca.it0 = it;
                ca.test_not_resident = (++launch_no == fail_launch) ? 1 : 0;
                HIPK_CHECK_HIP(hipMemsetAsync(ca.r_ll, 0, 2 * vec8, stream));
                HIPK_CHECK_HIP(hipMemsetAsync(ca.pap_ll, 0, kMidSlotBytes, stream));
                HIPK_CHECK_HIP(hipMemsetAsync(&scal->ctl, 0, sizeof(hipk_lds_ctl), stream));
                mid_kern<<<hipk_xcd_grid(mid_grid), mid_threads, lds, stream>>>(ca);   // hipk_xcd_chunk: padded to a multiple of 8
                HIPK_CHECK_HIP(hipGetLastError());
                HIPK_CHECK_HIP(hipMemcpyAsync(&hs0, scal, sizeof(hs0), hipMemcpyDeviceToHost, stream));
                HIPK_CHECK_HIP(hipStreamSynchronize(stream));
                if (hs0.ctl.redo < 0) {
                    if (hs0.ctl.redo == -3) {
                        hipk_set_error("hipk_cg_solve: a resident workgroup of the one-launch loop stopped arriving");
                        return HIPK_ERR_HIP;
                    }
                    if (!getenv("HIPK_TEST_LDS_NOT_RESIDENT")) mid_failed = true;   // not co-resident; nothing was modified
                    mid_loop = false;
                    break;
                }
                it = hs0.ctl.it_done;
                if (hs0.stop_it <= it || it >= maxiter) break;
            }
        }
    }
    // launch-bound systems with short rows: the whole loop in one launch (hipk_cg_solve_lds_kernel), bounded iterations per launch
    static bool lds_loop_failed = false;   // its workgroups once failed to meet (a shared device): do not wait for that verdict again
    // up to 64 workgroups (8 chunks) on ONE XCD, up to 512 (64 chunks, n <= 131072) spread over the chip, two per compute unit
    const bool lds_spread = kGmSub * gm.g > 64;
    // (measured per iteration, one launch vs three launches: 5.5 vs 16 us at 8 chunks, 10.6 vs 19.9 at 16, 12.3 vs 16.6 at 32,
    // 18.3 vs 18.3 at 64: the agent-scope hand-offs grow with the workgroup count -- taken up to 32 chunks, n <= 65536)
    bool lds_loop = gm.g <= 32 && !getenv("HIPK_CG_NO_SMALL") && gm.ch == HIPK_BASE_CHUNK && A->max_row_len <= kCgRowRegs &&
                    prm->profile == 0 && maxiter > 0 && kGmSub * gm.g <= (lds_spread ? 2 * A->n_cu : 2 * (A->n_cu / 8)) &&
                    !lds_loop_failed && !getenv("HIPK_CG_NO_LDS_LOOP") && !(lds_spread && getenv("HIPK_NO_LDS_SPREAD")) && !mid_loop;
    if (lds_loop) {
        bool local = !lds_spread && !getenv("HIPK_CG_LOOP_AGENT");
        const char *e = getenv("HIPK_CG_LAUNCH_ITS");
        hipk_cg_lds_args<T> ca;
        ca.n = n;
        ca.g = gm.g;
        ca.crow = A->crow;
        ca.col = A->col;
        ca.val = (const T *)A->val;
        ca.x = x;
        ca.r = r;
        ca.p = p;
        ca.Ap = Ap;
        ca.ctl = &scal->ctl;
        ca.gamma = scal->gamma;
        ca.atol2 = &scal->atol2;
        ca.stop_it = &scal->stop_it;
        ca.dinv = nullptr;
        ca.rz0_parts = nullptr;
        ca.rz_sub = nullptr;
        ca.tile_pp = A->tile_part;
        ca.rr_sub = part_b;
        ca.flag_a = (unsigned long long *)(part_c + 1024);   // 2 x 512 words
        ca.flag_b = ca.flag_a + kHoMaxWg;
        ca.spread = lds_spread ? 1 : 0;
        const int lgrid = lds_spread ? kGmSub * gm.g : 8 * kGmSub * gm.g;
        ca.maxiter = maxiter;
        ca.max_its = e ? atoll(e) : 16384;
        if (ca.max_its < 1) ca.max_its = 1;
        // tests: HIPK_TEST_LDS_NOT_RESIDENT=k makes the k-th launch of this solve report its workgroups as not co-resident
        const int fail_launch = getenv("HIPK_TEST_LDS_NOT_RESIDENT") ? (atoi(getenv("HIPK_TEST_LDS_NOT_RESIDENT")) > 1 ? atoi(getenv("HIPK_TEST_LDS_NOT_RESIDENT")) : 1) : 0;
        int launch_no = 0;
        hipk_cg_scal hs0;
        for (;;) {
            ca.it0 = it;
            ca.test_not_resident = (++launch_no == fail_launch) ? 1 : 0;
            HIPK_CHECK_HIP(hipMemsetAsync(ca.flag_a, 0, 2 * kHoMaxWg * sizeof(unsigned long long), stream));
            HIPK_CHECK_HIP(hipMemsetAsync(&scal->ctl, 0, sizeof(hipk_lds_ctl), stream));
            if (local)
                hipk_cg_solve_lds_kernel<T, true, false><<<lgrid, HIPK_THREADS, 0, stream>>>(ca);
            else
                hipk_cg_solve_lds_kernel<T, false, false><<<lgrid, HIPK_THREADS, 0, stream>>>(ca);
            HIPK_CHECK_HIP(hipGetLastError());
            HIPK_CHECK_HIP(hipMemcpyAsync(&hs0, scal, sizeof(hs0), hipMemcpyDeviceToHost, stream));
            HIPK_CHECK_HIP(hipStreamSynchronize(stream));
            if (hs0.ctl.redo < 0) {
                if (hs0.ctl.redo == -3) {
                    hipk_set_error("hipk_cg_solve: a resident workgroup of the one-launch loop stopped arriving");
                    return HIPK_ERR_HIP;
                }
                if (hs0.ctl.redo == -2 && local) {   // spread over several XCDs: agent-scope hand-offs
                    local = false;
                    continue;
                }
                if (!getenv("HIPK_TEST_LDS_NOT_RESIDENT")) lds_loop_failed = true;   // not co-resident; nothing was modified: the launch sequence below takes over
                lds_loop = false;
                break;
            }
            it = hs0.ctl.it_done;
            if (hs0.stop_it <= it || it >= maxiter) break;
        }
    }
    if (mid_loop) lds_loop = true;   // finished in the one-launch loop: none of the launch sequences below runs
    // launch-bound mid-size systems: TWO launches per iteration (hipk_cg2_spmv_kernel / hipk_cg2_update_kernel above)
    constexpr int kCap2 = sizeof(T) == 8 ? 1280 : 2048;
    const bool two_launch = !lds_loop && !small && gm.g > 32 && gm.g <= kCg2MaxChunks && gm.ch == HIPK_BASE_CHUNK && A->op_cb == nullptr &&
                            A->crow != nullptr && A->max_tile_nnz <= kCap2 && A->max_row_len <= HIPK_LONG_ROW && prm->profile == 0 &&
                            it == 0 && !(getenv("HIPK_CG_TWO_LAUNCH") && getenv("HIPK_CG_TWO_LAUNCH")[0] == '0') &&
                            hipk_cg_work_bytes(n, A->dtype) >= 256 + hipk_scratch_bytes() + 4 * vec;
    if (two_launch) {
        T *pbuf[2] = {p, (T *)((char *)Ap + vec)};   // p_0 = r_0 sits in pbuf[0] (start kernel); pass k reads pbuf[k & 1], writes the other
        hipk_cg2_args ca;
        ca.crow = A->crow;
        ca.col = A->col;
        ca.val = A->val;
        ca.n = n;
        ca.ch = gm.ch;
        ca.g = gm.g;
        ca.scal = scal;
        ca.maxiter = maxiter;
        ca.part_rr = part_b;
        ca.r = r;
        ca.Ap = Ap;
        ca.tpart = A->tile_part;
        const int grid1 = ((ntiles + 7) >> 3) << 3;
        // pass `maxiter` is bookkeeping only (its K1 folds the last <r,r>, sets the stop word and gamma: TSL:841 "k >= maxiter")
        for (; it <= maxiter; ++it) {
            HIPK_CHECK_HIP(pace.gate(it, stream, &stop));
            if (stop <= it) break;
            ca.it = it;
            ca.p_old = pbuf[it & 1];
            ca.p_new = pbuf[(it + 1) & 1];
            hipk_cg2_spmv_kernel<T, kCap2><<<grid1, HIPK_THREADS, 0, stream>>>(ca);
            if (it < maxiter)
                hipk_cg2_update_kernel<T><<<gm.g, HIPK_THREADS, 0, stream>>>(n, gm.ch, gm.g, scal, it, A->tile_part, ntiles, Ap,
                                                                            (const T *)pbuf[(it + 1) & 1], r, x, part_b);
            if ((it & 63) == 63) HIPK_CHECK_HIP(hipGetLastError());
        }
        if (it > maxiter) it = maxiter;
    }
    for (; !lds_loop && !two_launch && it < maxiter; ++it) {
        HIPK_CHECK_HIP(pace.gate(it, stream, &stop));
        if (stop <= it) break;
        {
            sa.it = it;
            // params.profile selects the kernel whose durations are reported (1 SpMV, 2 update, 3 direction, 4 scalars)
            if ((rc = hipk_launch_spmv(A, sa, stream, &prof)) != HIPK_OK) return rc;
            if (small)
                hipk_launch_timed(&prof, HIPK_K_UPDATE, hipk_cg_update_kernel<T, true>, gm.g, HIPK_THREADS, 0, stream, n, gm.ch, gm.g, scal, it,
                                  A->tile_part, Ap, r, part_b, ntiles);
            else if (streams)
                hipk_launch_timed(&prof, HIPK_K_UPDATE, hipk_cg_update_kernel<T, false, true>, gm.g, HIPK_THREADS, 0, stream, n, gm.ch, gm.g, scal, it,
                                  part_a, Ap, r, part_b, 0);
            else
                hipk_launch_timed(&prof, HIPK_K_UPDATE, hipk_cg_update_kernel<T>, gm.g, HIPK_THREADS, 0, stream, n, gm.ch, gm.g, scal, it, part_a, Ap, r,
                                  part_b, 0);
            if (small)
                hipk_launch_timed(&prof, HIPK_K_DIRECTION, hipk_cg_direction_kernel<T, true>, gm.g, HIPK_THREADS, 0, stream, n, gm.ch, gm.g, scal, it,
                                  maxiter, A->tile_part, part_b, r, p, x, ntiles);
            else if (streams && flat_dir) {
                // profile 3 times the flat kernel (the step's 40 n bytes), profile 4 the scalars launch before it
                hipk_launch_timed(&prof, HIPK_K_SCALARS, hipk_cg_scalars_kernel, 1, HIPK_THREADS, 0, stream, gm.g, scal, it,
                                  maxiter, part_a, part_b);
                hipk_launch_timed(&prof, HIPK_K_DIRECTION, hipk_cg_direction_flat_kernel<T>, (unsigned)((n + HIPK_BASE_CHUNK - 1) / HIPK_BASE_CHUNK),
                                  HIPK_THREADS, 0, stream, n, scal, it, r, p, x);
            } else if (streams)
                hipk_launch_timed(&prof, HIPK_K_DIRECTION, hipk_cg_direction_kernel<T, false, true>, gm.g, HIPK_THREADS, 0, stream, n, gm.ch, gm.g, scal, it,
                                  maxiter, part_a, part_b, r, p, x, 0);
            else
                hipk_launch_timed(&prof, HIPK_K_DIRECTION, hipk_cg_direction_kernel<T>, gm.g, HIPK_THREADS, 0, stream, n, gm.ch, gm.g, scal, it, maxiter,
                                  part_a, part_b, r, p, x, 0);
        }
        if ((it & 63) == 63) HIPK_CHECK_HIP(hipGetLastError());
    }
    HIPK_CHECK_HIP(hipGetLastError());

    // ---- TSL:1007-1014: true residual, ||x||
    sa.x = x;
    sa.y = Ap;
    sa.mode = HIPK_SPMV_RESID | HIPK_SPMV_DOT_YY;
    sa.w = nullptr;
    sa.bsub = b;
    sa.part0 = part_c;
    sa.part1 = part_b;
    sa.stop_it = nullptr;
    sa.skip_combine = 0;
    if ((rc = hipk_launch_spmv(A, sa, stream)) != HIPK_OK) return rc;
    ++matvecs;
    if ((rc = hipk_launch_dot_parts(n, x, x, A->dtype, part_a, stream)) != HIPK_OK) return rc;
    hipk_cg_final_kernel<<<1, HIPK_THREADS, 0, stream>>>(scal, gm.g, part_b, part_a);
    HIPK_CHECK_HIP(hipGetLastError());
    hipk_cg_scal hs;
    HIPK_CHECK_HIP(hipEventRecord(whole.b, stream));
    HIPK_CHECK_HIP(hipMemcpyAsync(&hs, scal, sizeof(hs), hipMemcpyDeviceToHost, stream));
    HIPK_CHECK_HIP(hipStreamSynchronize(stream));

    const int64_t iterations = (hs.stop_it < it) ? hs.stop_it : it;  // the device's stop word is authoritative
    matvecs += iterations;
    hipk_finish_isolve_stats(st, prm, hs.bs, hs.res2, hs.xx, iterations, matvecs);
    st->recurrence_rs = hs.gamma[iterations & 1];
    st->breakdown = 0;
    float ms = 0.f;
    HIPK_CHECK_HIP(hipEventElapsedTime(&ms, whole.a, whole.b));
    st->solve_ms = ms;
    HIPK_CHECK_HIP(prof.collect(st, iterations));
    return HIPK_OK;
}

extern "C" int hipk_cg_solve(hipk_csr_t A, const void *b, void *x, void *work, size_t work_bytes,
                             const hipk_params *prm, hipk_stats *st, hipk_stream_t stream) {
    HIPK_REQUIRE(A && b && x && work && prm && st, HIPK_ERR_ARG, "null argument");
    HIPK_REQUIRE(A->n_rows == A->n_cols, HIPK_ERR_ARG, "linear operator must be a square matrix");
    HIPK_REQUIRE(A->n_rows > 0, HIPK_ERR_ARG, "empty system");
    HIPK_REQUIRE(hipk_aligned16(b) && hipk_aligned16(x) && (((uintptr_t)work) & 255u) == 0, HIPK_ERR_ALIGN,
                 "b/x must be 16-byte and work 256-byte aligned");
    HIPK_REQUIRE(work_bytes >= hipk_cg_work_bytes(A->n_rows, A->dtype), HIPK_ERR_WORKSPACE, "work too small");
    HIPK_REQUIRE(b != x, HIPK_ERR_ARG, "b and x must not alias");
    memset(st, 0, sizeof(*st));
    if (A->dtype == HIPK_F64)
        return hipk_cg_solve_t<double>(A, (const double *)b, (double *)x, (char *)work, prm, st, (hipStream_t)stream);
    return hipk_cg_solve_t<float>(A, (const float *)b, (float *)x, (char *)work, prm, st, (hipStream_t)stream);
}

// ------------------------------------------------------------------ step API (include/hipk.h)
extern "C" size_t hipk_cg_scal_bytes(void) { return sizeof(hipk_cg_scal); }

static int hipk_step_check(int64_t n_local, int chunk_rows, int g_red, int dtype) {
    HIPK_REQUIRE(n_local > 0, HIPK_ERR_ARG, "n_local must be positive");
    HIPK_REQUIRE(chunk_rows >= HIPK_BASE_CHUNK && chunk_rows % HIPK_BASE_CHUNK == 0, HIPK_ERR_ARG, "chunk_rows");
    HIPK_REQUIRE(g_red >= 1 && g_red <= HIPK_MAX_PARTS, HIPK_ERR_ARG, "g_red out of range");
    HIPK_REQUIRE((n_local + chunk_rows - 1) / chunk_rows <= g_red, HIPK_ERR_ARG, "more local chunks than g_red");
    HIPK_REQUIRE(dtype == HIPK_F64 || dtype == HIPK_F32, HIPK_ERR_UNSUPPORTED, "dtype");
    return HIPK_OK;
}

extern "C" int hipk_cg_start(int64_t n_local, int chunk_rows, int g_red, void *scal_dev, const double *part_rr,
                             const double *part_bb, const void *r, void *p, int dtype, double tol, double atol,
                             int64_t maxiter, hipk_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    int rc = hipk_step_check(n_local, chunk_rows, g_red, dtype);
    if (rc != HIPK_OK) return rc;
    HIPK_REQUIRE(scal_dev && part_rr && part_bb && r && p, HIPK_ERR_ARG, "null argument");
    const float tolf = (float)tol, atolf = (float)atol;
    const double tol2 = (double)(tolf * tolf), atol_sq = (double)(atolf * atolf);
    const int grid = (int)((n_local + chunk_rows - 1) / chunk_rows);
    if (dtype == HIPK_F64)
        hipk_cg_start_kernel<double><<<grid, HIPK_THREADS, 0, stream>>>(n_local, chunk_rows, g_red, (hipk_cg_scal *)scal_dev,
                                                                         part_rr, part_bb, (const double *)r, (double *)p,
                                                                         tol2, atol_sq, maxiter);
    else
        hipk_cg_start_kernel<float><<<grid, HIPK_THREADS, 0, stream>>>(n_local, chunk_rows, g_red, (hipk_cg_scal *)scal_dev,
                                                                        part_rr, part_bb, (const float *)r, (float *)p, tol2,
                                                                        atol_sq, maxiter);
    HIPK_CHECK_HIP(hipGetLastError());
    return HIPK_OK;
}

extern "C" int hipk_cg_update(int64_t n_local, int chunk_rows, int g_red, const void *scal_dev, int64_t it,
                              const double *part_pAp, const void *Ap, void *r, double *part_rr_out, int dtype,
                              hipk_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    int rc = hipk_step_check(n_local, chunk_rows, g_red, dtype);
    if (rc != HIPK_OK) return rc;
    HIPK_REQUIRE(scal_dev && part_pAp && Ap && r && part_rr_out, HIPK_ERR_ARG, "null argument");
    const int grid = (int)((n_local + chunk_rows - 1) / chunk_rows);
    if (dtype == HIPK_F64)
        hipk_cg_update_kernel<double><<<grid, HIPK_THREADS, 0, stream>>>(n_local, chunk_rows, g_red,
                                                                          (const hipk_cg_scal *)scal_dev, it, part_pAp,
                                                                          (const double *)Ap, (double *)r, part_rr_out);
    else
        hipk_cg_update_kernel<float><<<grid, HIPK_THREADS, 0, stream>>>(n_local, chunk_rows, g_red,
                                                                         (const hipk_cg_scal *)scal_dev, it, part_pAp,
                                                                         (const float *)Ap, (float *)r, part_rr_out);
    HIPK_CHECK_HIP(hipGetLastError());
    return HIPK_OK;
}

extern "C" int hipk_cg_direction(int64_t n_local, int chunk_rows, int g_red, void *scal_dev, int64_t it,
                                 int64_t maxiter, const double *part_pAp, const double *part_rr, const void *r, void *p,
                                 void *x, int dtype, hipk_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    int rc = hipk_step_check(n_local, chunk_rows, g_red, dtype);
    if (rc != HIPK_OK) return rc;
    HIPK_REQUIRE(scal_dev && part_pAp && part_rr && r && p, HIPK_ERR_ARG, "null argument");   // x == NULL: p only (see hipk_cg_xupdate)
    const int grid = (int)((n_local + chunk_rows - 1) / chunk_rows);
    if (dtype == HIPK_F64 && x)
        hipk_cg_direction_kernel<double><<<grid, HIPK_THREADS, 0, stream>>>(n_local, chunk_rows, g_red,
                                                                             (hipk_cg_scal *)scal_dev, it, maxiter,
                                                                             part_pAp, part_rr, (const double *)r,
                                                                             (double *)p, (double *)x);
    else if (dtype == HIPK_F64)
        hipk_cg_direction_kernel<double, false, false, true><<<grid, HIPK_THREADS, 0, stream>>>(
            n_local, chunk_rows, g_red, (hipk_cg_scal *)scal_dev, it, maxiter, part_pAp, part_rr, (const double *)r, (double *)p, nullptr);
    else if (x)
        hipk_cg_direction_kernel<float><<<grid, HIPK_THREADS, 0, stream>>>(n_local, chunk_rows, g_red,
                                                                            (hipk_cg_scal *)scal_dev, it, maxiter,
                                                                            part_pAp, part_rr, (const float *)r,
                                                                            (float *)p, (float *)x);
    else
        hipk_cg_direction_kernel<float, false, false, true><<<grid, HIPK_THREADS, 0, stream>>>(
            n_local, chunk_rows, g_red, (hipk_cg_scal *)scal_dev, it, maxiter, part_pAp, part_rr, (const float *)r, (float *)p, nullptr);
    HIPK_CHECK_HIP(hipGetLastError());
    return HIPK_OK;
}

extern "C" int hipk_cg_xupdate(int64_t n_local, int chunk_rows, int g_red, const void *scal_dev, int64_t it,
                               const double *part_pAp, const void *p, void *x, int dtype, hipk_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    int rc = hipk_step_check(n_local, chunk_rows, g_red, dtype);
    if (rc != HIPK_OK) return rc;
    HIPK_REQUIRE(scal_dev && part_pAp && p && x, HIPK_ERR_ARG, "null argument");
    const int grid = (int)((n_local + chunk_rows - 1) / chunk_rows);
    if (dtype == HIPK_F64)
        hipk_cg_xupdate_kernel<double><<<grid, HIPK_THREADS, 0, stream>>>(n_local, chunk_rows, g_red, (const hipk_cg_scal *)scal_dev, it,
                                                                          part_pAp, (const double *)p, (double *)x);
    else
        hipk_cg_xupdate_kernel<float><<<grid, HIPK_THREADS, 0, stream>>>(n_local, chunk_rows, g_red, (const hipk_cg_scal *)scal_dev, it,
                                                                         part_pAp, (const float *)p, (float *)x);
    HIPK_CHECK_HIP(hipGetLastError());
    return HIPK_OK;
}

// ---- step API for CG with a CALLABLE preconditioner ("arbitrary callable M between fused kernels") ----
// The host drives  SpMV+<p,Ap> | hipk_cg_update (r, <r,r>) | z = M(r) by the caller, any device code on the same
// stream | hipk_dot_parts(r, z) | hipk_cgm_direction.  gamma = <r,z> steers alpha and beta, the stop test uses
// rs = <r,r> (TSL:835-841); the arithmetic per element is hipk_pcg_direction_kernel's with z read instead of formed,
// so M = (r -> dinv * r) reproduces hipk_pcg_solve bit for bit.
template <typename T>
__global__ __launch_bounds__(HIPK_THREADS) void hipk_cgm_start_kernel(
    int64_t n, int ch, int g, hipk_cg_scal *__restrict__ scal, const double *__restrict__ part_rz,
    const double *__restrict__ part_rr, const double *__restrict__ part_bb, const T *__restrict__ z, T *__restrict__ p,
    double tol2, double atol_sq, int64_t maxiter) {
    __shared__ double sbuf[2 * HIPK_THREADS];
    double gamma0, rr0;
    hipk_reduce_parts2(part_rz, part_rr, g, gamma0, rr0, sbuf);
    const double bs = hipk_reduce_parts(part_bb, g, sbuf);
    hipk_chunk_loop<T>(n, ch, blockIdx.x, [&](int64_t i, int nv) {
        T zv[hipk_vec<T>::VEC];
        hipk_ld<T>(z, i, nv, zv);
        hipk_st<T>(p, i, nv, zv);  // p0 = z0 (TSL:822)
    });
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const double a2 = tol2 * bs;
        const double atol2 = (a2 > atol_sq) ? a2 : atol_sq;
        scal->gamma[0] = gamma0;
        scal->gamma[1] = 0.0;
        scal->atol2 = atol2;
        scal->bs = bs;
        scal->stop_it = (maxiter <= 0 || rr0 <= atol2) ? 0 : INT64_MAX;  // TSL:841 before the first SpMV
        scal->host_sig = nullptr;
    }
}

template <typename T>
__global__ __launch_bounds__(HIPK_THREADS) void hipk_cgm_direction_kernel(
    int64_t n, int ch, int g, hipk_cg_scal *__restrict__ scal, int64_t it, int64_t maxiter,
    const double *__restrict__ part_pAp, const double *__restrict__ part_rz, const double *__restrict__ part_rr,
    const T *__restrict__ z, T *__restrict__ p, T *__restrict__ x) {
    const int c = blockIdx.x;
    hipk_pre<T, 2> pre;
    pre.issue(n, ch, c, {z, (const T *)p});
    if (it >= scal->stop_it) return;
    __shared__ double sbuf[2 * HIPK_THREADS];
    double pAp, gamma_new;
    hipk_reduce_parts2(part_pAp, part_rz, g, pAp, gamma_new, sbuf);
    const double rr = hipk_reduce_parts(part_rr, g, sbuf);
    const double gamma = scal->gamma[it & 1];
    const T alpha = (T)(gamma / pAp);       // the bits hipk_cg_update_kernel derived
    const T beta = (T)(gamma_new / gamma);  // TSL:851
    pre.run([&](int64_t i, int nv, T(&v)[2][hipk_vec<T>::VEC]) {
        constexpr int VEC = hipk_vec<T>::VEC;
        T xv[VEC], pv[VEC];
        hipk_ld<T>((const T *)x, i, nv, xv);
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            const T m0 = alpha * v[1][k];
            xv[k] = xv[k] + m0;  // TSL:847
            const T m = beta * v[1][k];
            pv[k] = v[0][k] + m;  // TSL:852
        }
        hipk_st<T>(x, i, nv, xv);
        hipk_st<T>(p, i, nv, pv);
    });
    if (c == 0 && threadIdx.x == 0) {
        scal->gamma[(it + 1) & 1] = gamma_new;
        if (it + 1 >= maxiter || rr <= scal->atol2) scal->stop_it = it + 1;  // TSL:841 for the next pass
    }
}

extern "C" int hipk_cgm_start(int64_t n_local, int chunk_rows, int g_red, void *scal_dev, const double *part_rz,
                              const double *part_rr, const double *part_bb, const void *z, void *p, int dtype,
                              double tol, double atol, int64_t maxiter, hipk_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    int rc = hipk_step_check(n_local, chunk_rows, g_red, dtype);
    if (rc != HIPK_OK) return rc;
    HIPK_REQUIRE(scal_dev && part_rz && part_rr && part_bb && z && p, HIPK_ERR_ARG, "null argument");
    const float tolf = (float)tol, atolf = (float)atol;
    const double tol2 = (double)(tolf * tolf), atol_sq = (double)(atolf * atolf);
    const int grid = (int)((n_local + chunk_rows - 1) / chunk_rows);
    if (dtype == HIPK_F64)
        hipk_cgm_start_kernel<double><<<grid, HIPK_THREADS, 0, stream>>>(n_local, chunk_rows, g_red, (hipk_cg_scal *)scal_dev,
                                                                          part_rz, part_rr, part_bb, (const double *)z,
                                                                          (double *)p, tol2, atol_sq, maxiter);
    else
        hipk_cgm_start_kernel<float><<<grid, HIPK_THREADS, 0, stream>>>(n_local, chunk_rows, g_red, (hipk_cg_scal *)scal_dev,
                                                                         part_rz, part_rr, part_bb, (const float *)z,
                                                                         (float *)p, tol2, atol_sq, maxiter);
    HIPK_CHECK_HIP(hipGetLastError());
    return HIPK_OK;
}

extern "C" int hipk_cgm_direction(int64_t n_local, int chunk_rows, int g_red, void *scal_dev, int64_t it, int64_t maxiter,
                                  const double *part_pAp, const double *part_rz, const double *part_rr, const void *z,
                                  void *p, void *x, int dtype, hipk_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    int rc = hipk_step_check(n_local, chunk_rows, g_red, dtype);
    if (rc != HIPK_OK) return rc;
    HIPK_REQUIRE(scal_dev && part_pAp && part_rz && part_rr && z && p && x, HIPK_ERR_ARG, "null argument");
    const int grid = (int)((n_local + chunk_rows - 1) / chunk_rows);
    if (dtype == HIPK_F64)
        hipk_cgm_direction_kernel<double><<<grid, HIPK_THREADS, 0, stream>>>(
            n_local, chunk_rows, g_red, (hipk_cg_scal *)scal_dev, it, maxiter, part_pAp, part_rz, part_rr, (const double *)z,
            (double *)p, (double *)x);
    else
        hipk_cgm_direction_kernel<float><<<grid, HIPK_THREADS, 0, stream>>>(
            n_local, chunk_rows, g_red, (hipk_cg_scal *)scal_dev, it, maxiter, part_pAp, part_rz, part_rr, (const float *)z,
            (float *)p, (float *)x);
    HIPK_CHECK_HIP(hipGetLastError());
    return HIPK_OK;
}

// =====================================================================================================
// CG with a Jacobi preconditioner, M = diag(dinv)  (TSL:806-856 with `M is not _identity`).
// z = M r is never stored: the update kernel forms it for <r,z>, the direction kernel forms it again (same
// operands, same bits) for p = z + beta p.  gamma = <r,z> drives alpha and beta; the stop test uses rs = <r,r>
// (TSL:835-841); the final `info` compares ||M (b - A x)|| (TSL:1007).  Per iteration: SpMV + 32 n + 48 n bytes
// (16 n more than plain CG: dinv is read by both vector kernels).  Mirrored by orc_pcg_jacobi.
//   partial slots: a <p,Ap> | b <r,r> | c spare dot of the SpMV | z0, z1 <r,z> ping-pong (read by all workgroups
//   of the update kernel while the early ones already write the next one) | d <b,b> / <x,x>
static constexpr int kPcgMidMaxChunks = 256;                                 // one chunk per workgroup, one workgroup per CU
static constexpr size_t kPcgMidSlotBytes = 3 * (size_t)kMidMaxChunks * 256;   // <p,Ap>, <r,r>, <r,z> slot arrays
struct hipk_pcg_scal {
    double atol2, bs, res2, xx, rs_last;
    int64_t stop_it;
    int64_t *host_sig;  // as in hipk_cg_scal
    int64_t pad;
    double gamma[2];    // hipk_cg_solve_lds_kernel<.., PRE>: <r,z> by iteration parity (the launch sequence keeps it as partials)
    hipk_lds_ctl ctl;
};
static_assert(sizeof(hipk_pcg_scal) <= 256, "the scalar block is 256 bytes");

template <typename T>
__global__ __launch_bounds__(HIPK_THREADS) void hipk_pcg_start_kernel(
    int64_t n, int ch, int g, hipk_pcg_scal *__restrict__ scal, const double *__restrict__ part_rr,
    const double *__restrict__ part_bb, const T *__restrict__ r, const T *__restrict__ dinv, T *__restrict__ p,
    double *__restrict__ part_rz, double tol2, double atol_sq, int64_t maxiter, int64_t *host_sig) {
    __shared__ double sbuf[2 * HIPK_THREADS];
    double rr0, bs;
    hipk_reduce_parts2(part_rr, part_bb, g, rr0, bs, sbuf);
    const int c = blockIdx.x;
    double acc = 0.0;
    hipk_chunk_loop<T>(n, ch, c, [&](int64_t i, int nv) {
        constexpr int VEC = hipk_vec<T>::VEC;
        T rv[VEC], dv[VEC], zv[VEC];
        hipk_ld<T>(r, i, nv, rv);
        hipk_ld<T>(dinv, i, nv, dv);
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            zv[k] = dv[k] * rv[k];  // z0 = M r0 (TSL:821)
            if (k < nv) acc = fma((double)rv[k], (double)zv[k], acc);  // gamma0 = <r0, z0> (TSL:826)
        }
        hipk_st<T>(p, i, nv, zv);  // p0 = z0
    });
    acc = hipk_block_sum(acc, sbuf);
    if (threadIdx.x == 0) part_rz[c] = acc;
    if (c == 0 && threadIdx.x == 0) {
        const double a2 = tol2 * bs;
        const double atol2 = (a2 > atol_sq) ? a2 : atol_sq;
        scal->atol2 = atol2;
        scal->bs = bs;
        scal->rs_last = rr0;
        const bool done = (maxiter <= 0 || rr0 <= atol2);  // TSL:841 before the first SpMV
        scal->stop_it = done ? 0 : INT64_MAX;
        scal->host_sig = host_sig;
        if (done) hipk_signal(host_sig, HIPK_SIG_STOP);
    }
}

template <typename T>
__global__ __launch_bounds__(HIPK_THREADS) void hipk_pcg_update_kernel(
    int64_t n, int ch, int g, const hipk_pcg_scal *__restrict__ scal, int64_t it, const double *__restrict__ part_pAp,
    const double *__restrict__ part_rz_in, const T *__restrict__ Ap, const T *__restrict__ dinv, T *__restrict__ r,
    double *__restrict__ part_rr, double *__restrict__ part_rz_out) {
    const int c = blockIdx.x;
    hipk_pre<T, 2> pre;
    pre.issue(n, ch, c, {Ap, (const T *)r});
    if (it >= scal->stop_it) return;
    __shared__ double sbuf[2 * HIPK_THREADS];
    double pAp, gamma;
    hipk_reduce_parts2(part_pAp, part_rz_in, g, pAp, gamma, sbuf);
    const T alpha = (T)(gamma / pAp);  // TSL:846
    double acc0 = 0.0, acc1 = 0.0;
    pre.run([&](int64_t i, int nv, T(&v)[2][hipk_vec<T>::VEC]) {
        constexpr int VEC = hipk_vec<T>::VEC;
        T rv[VEC], dv[VEC];
        hipk_ld<T>(dinv, i, nv, dv);
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            const T m1 = alpha * v[0][k];
            rv[k] = v[1][k] - m1;        // TSL:848
            const T z = dv[k] * rv[k];   // TSL:849
            if (k < nv) {
                acc0 = fma((double)rv[k], (double)rv[k], acc0);  // rs of the next test (TSL:838)
                acc1 = fma((double)rv[k], (double)z, acc1);      // TSL:850
            }
        }
        hipk_st<T>(r, i, nv, rv);
    });
    hipk_block_sum2(acc0, acc1, sbuf);
    if (threadIdx.x == 0) {
        part_rr[c] = acc0;
        part_rz_out[c] = acc1;
    }
}

template <typename T>
__global__ __launch_bounds__(HIPK_THREADS) void hipk_pcg_direction_kernel(
    int64_t n, int ch, int g, hipk_pcg_scal *__restrict__ scal, int64_t it, int64_t maxiter,
    const double *__restrict__ part_pAp, const double *__restrict__ part_rz_old, const double *__restrict__ part_rz_new,
    const double *__restrict__ part_rr, const T *__restrict__ r, const T *__restrict__ dinv, T *__restrict__ p,
    T *__restrict__ x) {
    const int c = blockIdx.x;
    hipk_pre<T, 1> pre;  // four operand streams: one early operand keeps the kernel at 8 workgroups per CU
    pre.issue(n, ch, c, {(const T *)p});
    if (it >= scal->stop_it) return;
    __shared__ double sbuf[2 * HIPK_THREADS];
    double pAp, gamma, gamma_new, rr;
    hipk_reduce_parts2(part_pAp, part_rz_old, g, pAp, gamma, sbuf);
    hipk_reduce_parts2(part_rz_new, part_rr, g, gamma_new, rr, sbuf);
    const T alpha = (T)(gamma / pAp);       // the bits hipk_pcg_update_kernel derived
    const T beta = (T)(gamma_new / gamma);  // TSL:851
    pre.run([&](int64_t i, int nv, T(&v)[1][hipk_vec<T>::VEC]) {
        constexpr int VEC = hipk_vec<T>::VEC;
        T rv[VEC], xv[VEC], dv[VEC], pv[VEC];
        hipk_ld<T>(r, i, nv, rv);
        hipk_ld<T>((const T *)x, i, nv, xv);
        hipk_ld<T>(dinv, i, nv, dv);
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            const T m0 = alpha * v[0][k];
            xv[k] = xv[k] + m0;              // TSL:847
            const T z = dv[k] * rv[k];       // TSL:849 again
            const T m = beta * v[0][k];
            pv[k] = z + m;                   // TSL:852
        }
        hipk_st<T>(x, i, nv, xv);
        hipk_st<T>(p, i, nv, pv);
    });
    if (c == 0 && threadIdx.x == 0) {
        scal->rs_last = rr;
        const bool done = (it + 1 >= maxiter || rr <= scal->atol2);  // TSL:841 for the next pass
        if (done) scal->stop_it = it + 1;
        hipk_signal(scal->host_sig, done ? (HIPK_SIG_STOP | (it + 1)) : (it + 1));
    }
}

// partials of || dinv .* res ||^2   (final `_norm(M(b - A x))`, TSL:1007)
template <typename T>
__global__ __launch_bounds__(HIPK_THREADS) void hipk_pcg_resnorm_kernel(int64_t n, int ch, const T *__restrict__ res,
                                                                        const T *__restrict__ dinv,
                                                                        double *__restrict__ part) {
    __shared__ double sbuf[HIPK_THREADS];
    const int c = blockIdx.x;
    double acc = 0.0;
    hipk_chunk_loop<T>(n, ch, c, [&](int64_t i, int nv) {
        constexpr int VEC = hipk_vec<T>::VEC;
        T rv[VEC], dv[VEC];
        hipk_ld<T>(res, i, nv, rv);
        hipk_ld<T>(dinv, i, nv, dv);
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            const T m = dv[k] * rv[k];
            if (k < nv) acc = fma((double)m, (double)m, acc);
        }
    });
    acc = hipk_block_sum(acc, sbuf);
    if (threadIdx.x == 0) part[c] = acc;
}

__global__ __launch_bounds__(HIPK_THREADS) void hipk_pcg_final_kernel(hipk_pcg_scal *__restrict__ scal, int g,
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

extern "C" size_t hipk_pcg_work_bytes(int64_t n, int dtype) {
    const size_t sv = (dtype == HIPK_F64) ? 8 : 4;
    const size_t vec = hipk_align_up((size_t)(n > 0 ? n : 1) * sv, 256);
    // mid-size systems (the one-launch loop, hipk_cg_mid.h<.., PRE>): r as 16-byte flagged words in Ap + a fourth vector, three slot arrays
    const hipk_geom gm = hipk_make_geom(n > 0 ? n : 1);
    const bool mid = gm.g > kMidMinChunks && gm.g <= kPcgMidMaxChunks;
    const size_t ll = hipk_align_up((size_t)(n > 0 ? n : 1) * 16, 256);   // r as 16-byte flagged words, from Ap on
    return 256 + 6 * HIPK_MAX_PARTS * sizeof(double) + 3 * vec + (mid ? (ll - vec) + kPcgMidSlotBytes : 0);  // scalars | six partial slots | r, p, Ap
}

template <typename T>
static int hipk_pcg_solve_t(hipk_csr_s *A, const T *dinv, const T *b, T *x, char *work, const hipk_params *prm,
                            hipk_stats *st, hipStream_t stream) {
    const int64_t n = A->n_rows;
    const hipk_geom gm = A->geom;
    const size_t vec = hipk_align_up((size_t)(n > 0 ? n : 1) * sizeof(T), 256);
    hipk_pcg_scal *scal = (hipk_pcg_scal *)work;
    double *parts = (double *)(work + 256);
    double *part_a = parts, *part_b = parts + HIPK_MAX_PARTS, *part_c = parts + 2 * HIPK_MAX_PARTS;
    double *part_z[2] = {parts + 3 * HIPK_MAX_PARTS, parts + 4 * HIPK_MAX_PARTS};
    double *part_d = parts + 5 * HIPK_MAX_PARTS;
    T *r = (T *)(work + 256 + 6 * HIPK_MAX_PARTS * sizeof(double));
    T *p = (T *)((char *)r + vec);
    T *Ap = (T *)((char *)p + vec);

    const int64_t maxiter = (prm->maxiter < 0) ? 10 * n : prm->maxiter;
    const float tolf = (float)prm->tol, atolf = (float)prm->atol;
    const double tol2 = (double)(tolf * tolf), atol_sq = (double)(atolf * atolf);
    const int64_t check = prm->check_every > 0 ? prm->check_every : 64;

    hipk_event_pair whole;
    HIPK_CHECK_HIP(whole.create());
    HIPK_CHECK_HIP(hipEventRecord(whole.a, stream));

    hipk_spmv_args sa;
    memset(&sa, 0, sizeof(sa));
    sa.crow = A->crow;
    sa.col = A->col;
    sa.val = A->val;
    sa.n = n;
    sa.ch = gm.ch;
    sa.g = gm.g;
    int rc;
    int64_t matvecs = 0;

    // r0 = b - A x0 with <r0,r0> partials; <b,b>; z0, p0, gamma0 partials
    sa.x = x;
    sa.y = r;
    sa.mode = HIPK_SPMV_RESID | HIPK_SPMV_DOT_YY;
    sa.bsub = b;
    sa.part0 = part_c;
    sa.part1 = part_b;
    if ((rc = hipk_launch_spmv(A, sa, stream)) != HIPK_OK) return rc;
    ++matvecs;
    if ((rc = hipk_launch_dot_parts(n, b, b, A->dtype, part_d, stream)) != HIPK_OK) return rc;
    hipk_pacer pace(A->host_poll, &scal->stop_it, check);
    HIPK_CHECK_HIP(pace.create());
    hipk_pcg_start_kernel<T><<<gm.g, HIPK_THREADS, 0, stream>>>(n, gm.ch, gm.g, scal, part_b, part_d, r, dinv, p, part_z[0],
                                                                 tol2, atol_sq, maxiter, pace.device_sig());
    HIPK_CHECK_HIP(hipGetLastError());

    sa.x = p;
    sa.y = Ap;
    sa.mode = HIPK_SPMV_DOT_W;
    sa.w = p;
    sa.bsub = nullptr;
    sa.part0 = part_a;
    sa.part1 = part_c;
    sa.stop_it = &scal->stop_it;

    int64_t it = 0, stop = INT64_MAX;
    // launch-bound systems of 9 .. 256 chunks (fp64, rows of <= 12 entries within a window around their chunk): the whole loop in one
    // launch, one workgroup per chunk (hipk_cg_mid_kernel<W, 1, PRE = true>); HIPK_CG_MID=0 leaves them to the paths below
    static bool mid_failed = false;
    bool mid_loop = false;
    {
        mid_loop = gm.g > kMidMinChunks && gm.g <= kPcgMidMaxChunks && gm.g <= A->n_cu && gm.ch == HIPK_BASE_CHUNK && A->op_cb == nullptr &&
                   A->crow != nullptr && A->max_row_len <= 12 && prm->profile == 0 && maxiter > 0 && !mid_failed &&
                   !(getenv("HIPK_CG_MID") && getenv("HIPK_CG_MID")[0] == '0') && !getenv("HIPK_CG_NO_LDS_LOOP") && !getenv("HIPK_CG_NO_SMALL");
        void (*mid_kern)(hipk_cg_mid_args) = A->max_row_len <= 5   ? hipk_cg_mid_kernel<T, 5, 1, true>
                                             : A->max_row_len <= 7 ? hipk_cg_mid_kernel<T, 7, 1, true>
                                             : A->max_row_len <= 9 ? hipk_cg_mid_kernel<T, 9, 1, true>
                                                                   : hipk_cg_mid_kernel<T, 12, 1, true>;
        size_t lds = 0;
        hipk_mid_plan plan;
        memset(&plan, 0, sizeof(plan));
        if (mid_loop) {
            mid_loop = hipk_mid_plan_get(A, 1, stream, &plan);   // the tiles each workgroup's window holds (hipk_mid.h)
            lds = mid_loop ? hipk_cg_mid_lds_bytes(plan.max_slots * HIPK_TILE, 1, true, sizeof(T)) : 0;
            int occ = 0;
            mid_loop = mid_loop && plan.max_slots <= kMidPlanSlots && lds <= (size_t)160 * 1024 &&
                       hipFuncSetAttribute((const void *)mid_kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess &&
                       hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, mid_kern, 1024, lds) == hipSuccess && (int64_t)occ * A->n_cu >= gm.g;
            (void)hipGetLastError();
        }
        if (mid_loop) {
            const char *e = getenv("HIPK_CG_LAUNCH_ITS");
            hipk_cg_mid_args ca;
            memset(&ca, 0, sizeof(ca));
            ca.n = n;
            ca.g = gm.g;
            ca.win = plan.max_slots * HIPK_TILE;
            ca.plan = plan;
            ca.crow = A->crow;
            ca.col = A->col;
            ca.val = A->val;
            ca.x = x;
            ca.r = r;
            ca.p = p;
            ca.r_ll = (unsigned long long *)Ap;                          // Ap + the fourth vector: 2 x vec >= 16 n bytes
            const size_t ll_bytes = hipk_align_up((size_t)n * 16, 256);
            ca.pap_ll = (unsigned long long *)((char *)Ap + ll_bytes);    // behind the flagged words of r (hipk_pcg_work_bytes)
            ca.rr_ll = ca.pap_ll + (size_t)kMidMaxChunks * 256 / 8;
            ca.rz_ll = ca.rr_ll + (size_t)kMidMaxChunks * 256 / 8;
            ca.dinv = dinv;
            ca.rz0_parts = part_z[0];
            ca.slot_stride = gm.g <= 32 ? 1 : 16;
            ca.xcd_aware = 1;
            ca.ctl = &scal->ctl;
            ca.gamma = scal->gamma;
            ca.atol2 = &scal->atol2;
            ca.stop_it = &scal->stop_it;
            ca.maxiter = maxiter;
            ca.max_its = e ? atoll(e) : 16384;
            if (ca.max_its < 1) ca.max_its = 1;
            const int fail_launch = getenv("HIPK_TEST_LDS_NOT_RESIDENT") ? (atoi(getenv("HIPK_TEST_LDS_NOT_RESIDENT")) > 1 ? atoi(getenv("HIPK_TEST_LDS_NOT_RESIDENT")) : 1) : 0;
            int launch_no = 0;
            hipk_pcg_scal hs0;
            for (;;) {
                ca.it0 = it;
                ca.test_not_resident = (++launch_no == fail_launch) ? 1 : 0;
                HIPK_CHECK_HIP(hipMemsetAsync(ca.r_ll, 0, ll_bytes, stream));
                HIPK_CHECK_HIP(hipMemsetAsync(ca.pap_ll, 0, kPcgMidSlotBytes, stream));
                HIPK_CHECK_HIP(hipMemsetAsync(&scal->ctl, 0, sizeof(hipk_lds_ctl), stream));
                mid_kern<<<hipk_xcd_grid(gm.g), 1024, lds, stream>>>(ca);
                HIPK_CHECK_HIP(hipGetLastError());
                HIPK_CHECK_HIP(hipMemcpyAsync(&hs0, scal, sizeof(hs0), hipMemcpyDeviceToHost, stream));
                HIPK_CHECK_HIP(hipStreamSynchronize(stream));
                if (hs0.ctl.redo < 0) {
                    if (hs0.ctl.redo == -3) {
                        hipk_set_error("hipk_pcg_solve: a resident workgroup of the one-launch loop stopped arriving");
                        return HIPK_ERR_HIP;
                    }
                    if (!getenv("HIPK_TEST_LDS_NOT_RESIDENT")) mid_failed = true;   // this launch modified nothing
                    if (it > 0) {   // as below: <r,z> lives in scal->gamma[it & 1]; the launch sequence folds it from part_z[it & 1]
                        HIPK_CHECK_HIP(hipMemsetAsync(part_z[it & 1], 0, (size_t)gm.g * sizeof(double), stream));
                        HIPK_CHECK_HIP(hipMemcpyAsync(part_z[it & 1], &scal->gamma[it & 1], sizeof(double), hipMemcpyDeviceToDevice, stream));
                    }
                    mid_loop = false;
                    break;
                }
                it = hs0.ctl.it_done;
                if (hs0.stop_it <= it || it >= maxiter) break;
            }
        }
    }
    // launch-bound systems with short rows: the whole loop in one launch (hipk_cg_solve_lds_kernel<.., PRE = true>)
    static bool lds_loop_failed = false;
    const bool lds_spread = kGmSub * gm.g > 64;
    bool lds_loop = gm.g <= 32 && gm.ch == HIPK_BASE_CHUNK && A->max_row_len <= kCgRowRegs && prm->profile == 0 && maxiter > 0 &&
                    kGmSub * gm.g <= (lds_spread ? 2 * A->n_cu : 2 * (A->n_cu / 8)) && !lds_loop_failed &&
                    !getenv("HIPK_CG_NO_SMALL") && !getenv("HIPK_CG_NO_LDS_LOOP") && !(lds_spread && getenv("HIPK_NO_LDS_SPREAD")) && !mid_loop;
    if (lds_loop) {
        bool local = !lds_spread && !getenv("HIPK_CG_LOOP_AGENT");
        const char *e = getenv("HIPK_CG_LAUNCH_ITS");
        hipk_cg_lds_args<T> ca;
        ca.n = n;
        ca.g = gm.g;
        ca.crow = A->crow;
        ca.col = A->col;
        ca.val = (const T *)A->val;
        ca.x = x;
        ca.r = r;
        ca.p = p;
        ca.Ap = Ap;
        ca.ctl = &scal->ctl;
        ca.gamma = scal->gamma;
        ca.atol2 = &scal->atol2;
        ca.stop_it = &scal->stop_it;
        ca.dinv = dinv;
        ca.rz0_parts = part_z[0];
        ca.rz_sub = part_z[1];
        ca.tile_pp = A->tile_part;
        ca.rr_sub = part_b;
        ca.flag_a = (unsigned long long *)(part_c + 1024);   // 2 x 512 words
        ca.flag_b = ca.flag_a + kHoMaxWg;
        ca.spread = lds_spread ? 1 : 0;
        ca.maxiter = maxiter;
        ca.max_its = e ? atoll(e) : 16384;
        if (ca.max_its < 1) ca.max_its = 1;
        // tests: HIPK_TEST_LDS_NOT_RESIDENT=k makes the k-th launch of this solve report its workgroups as not co-resident
        const int fail_launch = getenv("HIPK_TEST_LDS_NOT_RESIDENT") ? (atoi(getenv("HIPK_TEST_LDS_NOT_RESIDENT")) > 1 ? atoi(getenv("HIPK_TEST_LDS_NOT_RESIDENT")) : 1) : 0;
        int launch_no = 0;
        const int lgrid = lds_spread ? kGmSub * gm.g : 8 * kGmSub * gm.g;
        hipk_pcg_scal hs0;
        for (;;) {
            ca.it0 = it;
            ca.test_not_resident = (++launch_no == fail_launch) ? 1 : 0;
            HIPK_CHECK_HIP(hipMemsetAsync(ca.flag_a, 0, 2 * kHoMaxWg * sizeof(unsigned long long), stream));
            HIPK_CHECK_HIP(hipMemsetAsync(&scal->ctl, 0, sizeof(hipk_lds_ctl), stream));
            if (local)
                hipk_cg_solve_lds_kernel<T, true, true><<<lgrid, HIPK_THREADS, 0, stream>>>(ca);
            else
                hipk_cg_solve_lds_kernel<T, false, true><<<lgrid, HIPK_THREADS, 0, stream>>>(ca);
            HIPK_CHECK_HIP(hipGetLastError());
            HIPK_CHECK_HIP(hipMemcpyAsync(&hs0, scal, sizeof(hs0), hipMemcpyDeviceToHost, stream));
            HIPK_CHECK_HIP(hipStreamSynchronize(stream));
            if (hs0.ctl.redo < 0) {
                if (hs0.ctl.redo == -3) {
                    hipk_set_error("hipk_pcg_solve: a resident workgroup of the one-launch loop stopped arriving");
                    return HIPK_ERR_HIP;
                }
                if (hs0.ctl.redo == -2 && local) {   // spread over several XCDs: agent-scope hand-offs
                    local = false;
                    continue;
                }
                if (!getenv("HIPK_TEST_LDS_NOT_RESIDENT")) lds_loop_failed = true;   // this launch modified nothing: the launch sequence below takes over
                if (it > 0) {
                    // ... from iteration `it` of an EARLIER launch: x, r, p are in memory, but <r,z> only as scal->gamma[it & 1]
                    // (part_z[1] was the kernel's sub-partial scratch), while the launch sequence folds it from the chunk
                    // partials part_z[it & 1].  Rebuild that slot as {gamma, 0, 0, ...}: the fold of it is gamma, bit for bit
                    HIPK_CHECK_HIP(hipMemsetAsync(part_z[it & 1], 0, (size_t)gm.g * sizeof(double), stream));
                    HIPK_CHECK_HIP(hipMemcpyAsync(part_z[it & 1], &scal->gamma[it & 1], sizeof(double), hipMemcpyDeviceToDevice, stream));
                }
                lds_loop = false;
                break;
            }
            it = hs0.ctl.it_done;
            if (hs0.stop_it <= it || it >= maxiter) break;
        }
    }
    if (mid_loop) lds_loop = true;   // finished in the one-launch loop
    for (; !lds_loop && it < maxiter; ++it) {
        HIPK_CHECK_HIP(pace.gate(it, stream, &stop));
        if (stop <= it) break;
        {
            sa.it = it;
            if ((rc = hipk_launch_spmv(A, sa, stream)) != HIPK_OK) return rc;
            hipk_pcg_update_kernel<T><<<gm.g, HIPK_THREADS, 0, stream>>>(n, gm.ch, gm.g, scal, it, part_a, part_z[it & 1], Ap,
                                                                          dinv, r, part_b, part_z[(it + 1) & 1]);
            hipk_pcg_direction_kernel<T><<<gm.g, HIPK_THREADS, 0, stream>>>(n, gm.ch, gm.g, scal, it, maxiter, part_a,
                                                                             part_z[it & 1], part_z[(it + 1) & 1], part_b, r,
                                                                             dinv, p, x);
        }
        if ((it & 63) == 63) HIPK_CHECK_HIP(hipGetLastError());
    }
    HIPK_CHECK_HIP(hipGetLastError());

    // TSL:1007-1014 with M: ||M (b - A x)||, ||x||
    sa.x = x;
    sa.y = Ap;
    sa.mode = HIPK_SPMV_RESID;
    sa.w = nullptr;
    sa.bsub = b;
    sa.part0 = part_c;
    sa.part1 = part_c;
    sa.stop_it = nullptr;
    if ((rc = hipk_launch_spmv(A, sa, stream)) != HIPK_OK) return rc;
    ++matvecs;
    hipk_pcg_resnorm_kernel<T><<<gm.g, HIPK_THREADS, 0, stream>>>(n, gm.ch, Ap, dinv, part_b);
    if ((rc = hipk_launch_dot_parts(n, x, x, A->dtype, part_d, stream)) != HIPK_OK) return rc;
    hipk_pcg_final_kernel<<<1, HIPK_THREADS, 0, stream>>>(scal, gm.g, part_b, part_d);
    HIPK_CHECK_HIP(hipGetLastError());
    hipk_pcg_scal hs;
    HIPK_CHECK_HIP(hipEventRecord(whole.b, stream));
    HIPK_CHECK_HIP(hipMemcpyAsync(&hs, scal, sizeof(hs), hipMemcpyDeviceToHost, stream));
    HIPK_CHECK_HIP(hipStreamSynchronize(stream));

    const int64_t iterations = (hs.stop_it < it) ? hs.stop_it : it;
    matvecs += iterations;
    hipk_finish_isolve_stats(st, prm, hs.bs, hs.res2, hs.xx, iterations, matvecs);
    st->recurrence_rs = (lds_loop && iterations > 0) ? hs.ctl.rs_last : hs.rs_last;
    st->breakdown = 0;
    float ms = 0.f;
    HIPK_CHECK_HIP(hipEventElapsedTime(&ms, whole.a, whole.b));
    st->solve_ms = ms;
    return HIPK_OK;
}

extern "C" int hipk_pcg_solve(hipk_csr_t A, const void *dinv, const void *b, void *x, void *work, size_t work_bytes,
                              const hipk_params *prm, hipk_stats *st, hipk_stream_t stream) {
    HIPK_REQUIRE(A && dinv && b && x && work && prm && st, HIPK_ERR_ARG, "null argument");
    HIPK_REQUIRE(A->n_rows == A->n_cols, HIPK_ERR_ARG, "linear operator must be a square matrix");
    HIPK_REQUIRE(A->n_rows > 0, HIPK_ERR_ARG, "empty system");
    HIPK_REQUIRE(hipk_aligned16(b) && hipk_aligned16(x) && hipk_aligned16(dinv) && (((uintptr_t)work) & 255u) == 0,
                 HIPK_ERR_ALIGN, "b/x/dinv must be 16-byte and work 256-byte aligned");
    HIPK_REQUIRE(work_bytes >= hipk_pcg_work_bytes(A->n_rows, A->dtype), HIPK_ERR_WORKSPACE, "work too small");
    HIPK_REQUIRE(b != x, HIPK_ERR_ARG, "b and x must not alias");
    memset(st, 0, sizeof(*st));
    if (A->dtype == HIPK_F64)
        return hipk_pcg_solve_t<double>(A, (const double *)dinv, (const double *)b, (double *)x, (char *)work, prm, st,
                                        (hipStream_t)stream);
    return hipk_pcg_solve_t<float>(A, (const float *)dinv, (const float *)b, (float *)x, (char *)work, prm, st,
                                   (hipStream_t)stream);
}

#ifdef HIPK_GM_STAMPS
// diagnostic twin only: per-workgroup phase time sums of the last hipk_cg_mid_kernel launch (hipk_cg_mid.h)
extern "C" int hipk_debug_mid_stamps(unsigned long long *out, size_t count) {
    const size_t have = sizeof(hipk_mid_stamps) / sizeof(unsigned long long);
    HIPK_CHECK_HIP(hipDeviceSynchronize());
    HIPK_CHECK_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(hipk_mid_stamps), sizeof(unsigned long long) * (count < have ? count : have)));
    return HIPK_OK;
}
#endif
