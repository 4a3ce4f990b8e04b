// hipk_bicgstab.hip -- device-resident BiCGStab.
//
// Restates `_isolve(_bicgstab_solve)` (TSL:859-964, 968-1016) for M = identity as five
// kernels per iteration; every scalar test of the reference (convergence TSL:894-896,
// rho breakdown :902-904, alpha breakdown :913-915, early exit :920, omega guard
// :926-930, omega breakdown :934-936) is evaluated on the device, redundantly and
// identically by every workgroup, from the chunk partials of the previous kernel:
//   K1 direction  rs, rho' -> tests; beta; p = r + beta (p - omega q)          32 n bytes
//   K2 spmv+dot   q = A p, <rhat,q>                                           B_spmv + 8 n
//   K3 s-update   alpha' = rho'/<rhat,q> -> test; s = r - alpha' q; <s,s>      24 n
//   K4 spmv+dots  t = A s, <t,s>, <t,t>                                        B_spmv
//   K5 x/r-update omega' -> tests; x += alpha' p (+ omega' s); r = s (- omega' t);
//                 <r,r>, <rhat,r> for the next iteration                       56 n
// = 2 B_spmv + 120 n bytes per iteration.  The full-vector `torch.where` selects of the
// reference (TSL:944, 950) become a wave-uniform branch.
#include <math.h>
#include <stdlib.h>

#include "hipk_blas1.h"
#include "hipk_solve.h"
#include "hipk_spmv.h"
#include "hipk_handoff.h"

#define HIPK_EPS64 2.220446049250313e-16
#define HIPK_EPS32 1.1920928955078125e-07

struct hipk_bi_scal {
    double rho, alpha, omega;  // committed by K5 of the last completed iteration
    double rho_new, alpha_new; // handed from K3 to K5 of the same iteration
    double atol2, bs;
    double rs_last;
    double res2, xx;
    int64_t stop_it;
    int64_t iters;
    int32_t code;      // 0 / -10 / -11  (TSL:903, 914, 935)
    int32_t extra_mv;  // SpMVs run by an iteration that then broke down
    int64_t *host_sig; // pinned host word the loop reports to (hipk_pacer, hipk_solve.h), or null
    // hipk_bi_solve_lds_kernel (small systems: the whole loop in one launch)
    int64_t it_done;   // iterations finished when the launch returned
    int32_t redo;      // < 0: its resident workgroups did not all arrive / were spread over several XCDs (nothing was modified)
    int32_t bar;       // counter barrier of its placement check
    unsigned xcc_mask;
    unsigned pad;
};
static_assert(sizeof(hipk_bi_scal) <= 256, "the scalar block is 256 bytes");

template <typename T>
struct hipk_eps;
template <>
struct hipk_eps<double> {
    static constexpr double v = HIPK_EPS64;
};
template <>
struct hipk_eps<float> {
    static constexpr double v = HIPK_EPS32;
};

#include "hipk_bi_mid.h"   // one-launch loop for mid-size systems (uses hipk_bi_scal, HIPK_EPS64)

template <typename T>
__global__ __launch_bounds__(HIPK_THREADS) void hipk_bi_start_kernel(
    int64_t n, int ch, int g, hipk_bi_scal *__restrict__ scal, const double *__restrict__ part_rr,
    const double *__restrict__ part_bb, double *__restrict__ part_rhr, const T *__restrict__ r, T *__restrict__ rhat,
    T *__restrict__ p, T *__restrict__ q, double tol2, double atol_sq, int64_t maxiter, int64_t *host_sig) {
    __shared__ double sbuf[2 * HIPK_THREADS];
    double rr, bs;
    hipk_reduce_parts2(part_rr, part_bb, g, rr, bs, sbuf);
    const int c = blockIdx.x;
    hipk_chunk_loop<T>(n, ch, c, [&](int64_t i, int nv) {
        T rv[hipk_vec<T>::VEC];
        hipk_ld<T>(r, i, nv, rv);
        hipk_st<T>(rhat, i, nv, rv);  // TSL:876
        hipk_st<T>(p, i, nv, rv);     // TSL:890
        hipk_st<T>(q, i, nv, rv);
    });
    if (threadIdx.x == 0) {
        part_rhr[c] = part_rr[c];  // <rhat, r0> = <r0, r0>
        if (c == 0) {
            const double a2 = tol2 * bs;
            scal->rho = 1.0;
            scal->alpha = 1.0;
            scal->omega = 1.0;
            scal->atol2 = (a2 > atol_sq) ? a2 : atol_sq;
            scal->bs = bs;
            scal->rs_last = 0.0;
            scal->stop_it = (maxiter <= 0) ? 0 : INT64_MAX;
            scal->host_sig = host_sig;
            if (maxiter <= 0) hipk_signal(host_sig, HIPK_SIG_STOP);
            scal->iters = 0;
            scal->code = 0;
            scal->extra_mv = 0;
        }
    }
}

// PRE = true: Jacobi preconditioning, M = diag(dinv) applied BEFORE A (TSL:908, 922): the kernel that forms p also
// stores phat = dinv .* p (the SpMV's input), the one that forms s also stores shat, and x is advanced with phat / shat.
template <typename T, bool PRE>
__global__ __launch_bounds__(HIPK_THREADS) void hipk_bi_direction_kernel(
    int64_t n, int ch, int g, hipk_bi_scal *__restrict__ scal, int64_t it, const double *__restrict__ part_rr,
    const double *__restrict__ part_rhr, const T *__restrict__ r, const T *__restrict__ q, T *__restrict__ p,
    const T *__restrict__ dinv, T *__restrict__ phat) {
    hipk_pre<T, 2> pre;  // r and q travel while the stop word is read and the partials are folded
    pre.issue(n, ch, blockIdx.x, {r, q});
    if (it >= scal->stop_it) return;
    __shared__ double sbuf[2 * HIPK_THREADS];
    double rs, rho_new;
    hipk_reduce_parts2(part_rr, part_rhr, g, rs, rho_new, sbuf);
    const double rho = scal->rho, alpha = scal->alpha, omega = scal->omega;
    const bool lead = (blockIdx.x == 0 && threadIdx.x == 0);
    if (lead) scal->rs_last = rs;
    if (rs <= scal->atol2) {  // TSL:894-896
        if (lead) {
            scal->stop_it = it;
            hipk_signal(scal->host_sig, HIPK_SIG_STOP | it);
        }
        return;
    }
    if (fabs(rho_new) < hipk_eps<T>::v * fabs(rho)) {  // TSL:902-904
        if (lead) {
            scal->stop_it = it;
            scal->code = -10;
            hipk_signal(scal->host_sig, HIPK_SIG_STOP | it);
        }
        return;
    }
    const T beta = (T)(rho_new / rho * alpha / omega);  // TSL:906, left to right
    const T om = (T)omega;
    pre.run([&](int64_t i, int nv, T(&v)[2][hipk_vec<T>::VEC]) {
        constexpr int VEC = hipk_vec<T>::VEC;
        T pv[VEC];
        hipk_ld<T>((const T *)p, i, nv, pv);
#pragma unroll
        for (int k = 0; k < VEC; ++k) {  // TSL:907
            const T t1 = om * v[1][k];
            const T t2 = pv[k] - t1;
            const T t3 = beta * t2;
            pv[k] = v[0][k] + t3;
        }
        hipk_st<T>(p, i, nv, pv);
        if (PRE) {
            T dv[VEC];
            hipk_ld<T>(dinv, i, nv, dv);
#pragma unroll
            for (int k = 0; k < VEC; ++k) dv[k] = dv[k] * pv[k];  // TSL:908
            hipk_st<T>(phat, i, nv, dv);
        }
    });
}

template <typename T, bool PRE, bool SMALL>
__global__ __launch_bounds__(HIPK_THREADS) void hipk_bi_supdate_kernel(
    int64_t n, int ch, int g, hipk_bi_scal *__restrict__ scal, int64_t it, const double *__restrict__ part_rhr,
    const double *__restrict__ part_rq, const T *__restrict__ r, const T *__restrict__ q, T *__restrict__ s,
    double *__restrict__ part_ss, const T *__restrict__ dinv, T *__restrict__ shat, int small_ntiles) {
    hipk_pre<T, 2> pre;
    pre.issue(n, ch, blockIdx.x, {r, q});
    if (it >= scal->stop_it) return;
    __shared__ double sbuf[2 * HIPK_THREADS];
    double rho_new, rq;
    if (SMALL) {  // small systems: <rhat,q> from the SpMV's tile sums, no combine launch (hipk_fold_tiles8); a compile-time
                  // switch: the fold's registers would cost the large-system instantiation its 8 workgroups per CU
        rq = hipk_fold_tiles8(part_rq, small_ntiles, ch / HIPK_TILE, g, sbuf);
        rho_new = hipk_reduce_parts(part_rhr, g, sbuf);
    } else {
        hipk_reduce_parts2(part_rhr, part_rq, g, rho_new, rq, sbuf);
    }
    const double alpha_new = rho_new / rq;  // TSL:910
    const bool lead = (blockIdx.x == 0 && threadIdx.x == 0);
    if (fabs(alpha_new) < hipk_eps<T>::v) {  // TSL:913-915
        if (lead) {
            scal->stop_it = it;
            scal->code = -11;
            scal->extra_mv = 1;
            hipk_signal(scal->host_sig, HIPK_SIG_STOP | it);
        }
        return;
    }
    if (lead) {
        scal->rho_new = rho_new;
        scal->alpha_new = alpha_new;
    }
    const T al = (T)alpha_new;
    double acc = 0.0;
    pre.run([&](int64_t i, int nv, T(&v)[2][hipk_vec<T>::VEC]) {
        constexpr int VEC = hipk_vec<T>::VEC;
        T sv[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            const T m = al * v[1][k];
            sv[k] = v[0][k] - m;  // TSL:917
            if (k < nv) acc = fma((double)sv[k], (double)sv[k], acc);
        }
        hipk_st<T>(s, i, nv, sv);
        if (PRE) {
            T dv[VEC];
            hipk_ld<T>(dinv, i, nv, dv);
#pragma unroll
            for (int k = 0; k < VEC; ++k) dv[k] = dv[k] * sv[k];  // TSL:922
            hipk_st<T>(shat, i, nv, dv);
        }
    });
    acc = hipk_block_sum(acc, sbuf);
    if (threadIdx.x == 0) part_ss[blockIdx.x] = acc;
}

template <typename T, bool PRE, bool SMALL>
__global__ __launch_bounds__(HIPK_THREADS) void hipk_bi_xupdate_kernel(
    int64_t n, int ch, int g, hipk_bi_scal *__restrict__ scal, int64_t it, int64_t maxiter,
    const double *__restrict__ part_ss, const double *__restrict__ part_ts, const double *__restrict__ part_tt,
    const T *__restrict__ p, const T *__restrict__ s, const T *__restrict__ t, const T *__restrict__ rhat,
    T *__restrict__ x, T *__restrict__ r, double *__restrict__ part_rr, double *__restrict__ part_rhr,
    const T *__restrict__ shat, int small_ntiles) {  // PRE: `p` is phat here; shat = M s
    hipk_pre<T, 1> pre;  // s up front; p, x, rhat, t follow after the fold (two early operands already cost the
    pre.issue(n, ch, blockIdx.x, {s});  // kernel its 8 workgroups per CU: 71 VGPRs)
    if (it >= scal->stop_it) return;
    __shared__ double sbuf[2 * HIPK_THREADS];
    const double ss = hipk_reduce_parts(part_ss, g, sbuf);
    double ts, tt;
    if (SMALL) {  // part_ts / part_tt point at the two tile-sum arrays of the second SpMV
        ts = hipk_fold_tiles8(part_ts, small_ntiles, ch / HIPK_TILE, g, sbuf);
        tt = hipk_fold_tiles8(part_tt, small_ntiles, ch / HIPK_TILE, g, sbuf);
    } else {
        hipk_reduce_parts2(part_ts, part_tt, g, ts, tt, sbuf);
    }
    const double atol2 = scal->atol2;
    const double alpha_new = scal->alpha_new, rho_new = scal->rho_new;
    const bool exit_early = ss < atol2;                                      // TSL:920 (strict)
    const double omega_new = (fabs(tt) < hipk_eps<T>::v) ? 0.0 : ts / tt;    // TSL:926-930
    const bool lead = (blockIdx.x == 0 && threadIdx.x == 0);
    if (fabs(omega_new) < hipk_eps<T>::v && !exit_early) {                   // TSL:934-936
        if (lead) {
            scal->stop_it = it;
            scal->code = -11;
            scal->extra_mv = 2;
            hipk_signal(scal->host_sig, HIPK_SIG_STOP | it);
        }
        return;
    }
    const T al = (T)alpha_new, om = (T)omega_new;
    double acc0 = 0.0, acc1 = 0.0;
    pre.run([&](int64_t i, int nv, T(&v)[1][hipk_vec<T>::VEC]) {
        constexpr int VEC = hipk_vec<T>::VEC;
        T(&sv)[VEC] = v[0];
        T pv[VEC], xv[VEC], hv[VEC], rv[VEC];
        hipk_ld<T>(p, i, nv, pv);
        hipk_ld<T>((const T *)x, i, nv, xv);
        hipk_ld<T>(rhat, i, nv, hv);
        if (exit_early) {  // TSL:942-950 with exit_early true
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                const T m0 = al * pv[k];
                xv[k] = xv[k] + m0;
                rv[k] = sv[k];
            }
        } else {
            T tv[VEC], shv[VEC];
            hipk_ld<T>(t, i, nv, tv);
            if (PRE) hipk_ld<T>(shat, i, nv, shv);
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                const T m0 = al * pv[k];
                const T m1 = om * (PRE ? shv[k] : sv[k]);  // TSL:942: omega * shat
                const T m2 = m0 + m1;
                xv[k] = xv[k] + m2;
                const T m3 = om * tv[k];
                rv[k] = sv[k] - m3;
            }
        }
#pragma unroll
        for (int k = 0; k < VEC; ++k)
            if (k < nv) {
                acc0 = fma((double)rv[k], (double)rv[k], acc0);
                acc1 = fma((double)hv[k], (double)rv[k], acc1);
            }
        hipk_st<T>(x, i, nv, xv);
        hipk_st<T>(r, i, nv, rv);
    });
    hipk_block_sum2(acc0, acc1, sbuf);
    if (threadIdx.x == 0) {
        part_rr[blockIdx.x] = acc0;
        part_rhr[blockIdx.x] = acc1;
    }
    if (lead) {
        scal->rho = rho_new;
        scal->alpha = alpha_new;
        scal->omega = omega_new;
        scal->iters = it + 1;
        const bool done = (exit_early || it + 1 >= maxiter);  // TSL:961, loop bound :892
        if (done) scal->stop_it = it + 1;
        hipk_signal(scal->host_sig, done ? (HIPK_SIG_STOP | (it + 1)) : (it + 1));
    }
}

// =====================================================================================================================
// Small systems (<= 32 reduction chunks, rows of <= 12 entries; M = identity or Jacobi): THE WHOLE BiCGStab LOOP IN ONE LAUNCH -- the scheme
// of hipk_cg_solve_lds_kernel (csrc/hipk_cg.hip) with three hand-offs per iteration instead of five launches:
//   K1 (tests, beta, p) | K2 q = A p by TILE rows, wavefront sums of rhat .* q            -> hand-off: tile sums + q
//   K3 (alpha test, s, <s,s> sub-partial) | K4 t = A s by tile rows, sums of s .* t, t .* t -> hand-off: tile sums + t + <s,s>
//   K5 (omega tests, x, r, sub-partials of <r,r> and <rhat,r>)                              -> hand-off: those + r
// The SpMV role keeps p, q, r at the columns of its tile row in registers and advances p = r + beta (p - omega q) and
// s = r - alpha q itself from the gathered r and q (the owners' formulas on the owners' operands): neither p nor s is exchanged.
// Arithmetic and every scalar test (TSL:894-936, 961) as in the five kernels above, bit for bit.
template <typename T>
struct hipk_bi_lds_args {
    int64_t n;
    int g;
    const int *crow;
    const int *col;
    const T *val;
    T *x, *r, *p, *q, *t;
    const T *rhat;
    const T *dinv;                  // PRE: Jacobi preconditioning, M = diag(dinv) applied BEFORE A (TSL:908, 922)
    hipk_bi_scal *scal;
    double *tsum0, *tsum1, *tsum2;  // [ntiles * 4] wavefront sums of <rhat,q>, <t,t>, <s,t> (three arrays: a fast workgroup writes
                                    // the sums of the second SpMV while a slow one still folds those of the first)
    double *part_rr, *part_rhr;     // it = 0: chunk partials of <r0,r0> (twice); afterwards [8 g] sub-partials
    double *part_ss;                // [8 g] sub-partials of <s,s>
    unsigned long long *flag_a, *flag_b, *flag_c;   // [64] each, zeroed before the launch
    int64_t it0, maxiter, max_its;
    int test_not_resident;   // tests (HIPK_TEST_LDS_NOT_RESIDENT): report the placement check as failed
    int spread;              // more than 64 workgroups: one per block all over the chip (then LOCAL = false)
};
static constexpr int kBiRowRegs = 12;

template <typename T, bool LOCAL, bool PRE>
__global__ __launch_bounds__(HIPK_THREADS, 2) void hipk_bi_solve_lds_kernel(hipk_bi_lds_args<T> a) {
    constexpr int VEC = hipk_vec<T>::VEC;
    constexpr double EPS = hipk_eps<T>::v;
    int wg = blockIdx.x;                             // spread (more than 64 workgroups): one per block, anywhere on the chip
    if (!a.spread) {
        if (blockIdx.x & 7) return;                  // the working blocks share an XCD (dispatch is round-robin over 8)
        wg = blockIdx.x >> 3;
    }
    const int c = wg / kGmSub, s_ = wg % kGmSub;
    const int g = a.g, nwg = g * kGmSub;
    if (c >= g) return;
    hipk_bi_scal *scal = a.scal;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int u = tid & 31, e8 = tid >> 5;
    const int64_t n = a.n;
    const int64_t base = (int64_t)c * HIPK_BASE_CHUNK;
    const int64_t row = base + (int64_t)VEC * (s_ + kGmSub * u) + (int64_t)(e8 / VEC) * (VEC * HIPK_THREADS) + (e8 % VEC);
    const bool live = row < n;
    const int ntiles = (int)((n + HIPK_TILE - 1) / HIPK_TILE);
    const int tile = c * (HIPK_BASE_CHUNK / HIPK_TILE) + s_;
    const int64_t trow = (int64_t)tile * HIPK_TILE + tid;
    const bool tlive = trow < n;

    __shared__ T wl[HIPK_THREADS], wh[HIPK_THREADS];
    __shared__ double bc[4];
    __shared__ int fail;
    __shared__ unsigned long long res_lds;
    if (tid == 0) fail = 0;

    // vector role
    T x_own = live ? a.x[row] : (T)0, r_own = live ? a.r[row] : (T)0, p_own = live ? a.p[row] : (T)0;
    T q_own = live ? a.q[row] : (T)0;
    const T h_own = live ? a.rhat[row] : (T)0;
    wh[tid] = h_own;
    // SpMV role: the tile row's entries; p and q at its columns (the launches before this one left them in memory)
    int lo = 0, len = 0;
    if (tlive) {
        lo = a.crow[trow];
        len = a.crow[trow + 1] - lo;
    }
    unsigned cj[kBiRowRegs];
    T vj[kBiRowRegs], pc[kBiRowRegs], qc[kBiRowRegs];
#pragma unroll
    for (int j = 0; j < kBiRowRegs; ++j) {
        const int cc = (j < len) ? a.col[lo + j] : 0;
        cj[j] = (unsigned)cc * (unsigned)sizeof(T);
        vj[j] = (j < len) ? a.val[lo + j] : (T)0;
        pc[j] = (j < len) ? a.p[cc] : (T)0;
        qc[j] = (j < len) ? a.q[cc] : (T)0;
    }
    const T h_t = tlive ? a.rhat[trow] : (T)0;
    T d_own = (T)1, dc[kBiRowRegs];   // PRE: the diagonal of M at the own row and at the tile row's columns
#pragma unroll
    for (int j = 0; j < kBiRowRegs; ++j) dc[j] = (T)1;
    if (PRE) {
        d_own = live ? a.dinv[row] : (T)0;
#pragma unroll
        for (int j = 0; j < kBiRowRegs; ++j) dc[j] = (j < len) ? a.dinv[a.col[lo + j]] : (T)0;
    }
    int wmax = len < kBiRowRegs ? len : kBiRowRegs;
    for (int off = 32; off > 0; off >>= 1) {
        const int o = __shfl_xor(wmax, off);
        wmax = o > wmax ? o : wmax;
    }
    wmax = __builtin_amdgcn_readfirstlane(wmax);
    double rho = scal->rho, alpha = scal->alpha, omega = scal->omega;
    const double atol2 = scal->atol2;
    const int64_t stop0 = scal->stop_it;

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
#define HIPK_BI_HO(flags)                                                      \
    if (hipk_ho_sync<LOCAL>(flags, wg, nwg, ++seq, 0u, &res_lds) == ~0ull) {   \
        if (tid == 0) scal->redo = -3;                                         \
        return;                                                                \
    }
    auto fold_tiles = [&](const double *tp, int i8) {   // i8 = lane of a whole wavefront
        return hipk_fold_64x8(i8, g, [&](int ci, int tt) {
            const int tl = ci * (HIPK_BASE_CHUNK / HIPK_TILE) + tt;
            if (tl >= ntiles) return 0.0;
            const double *w4 = tp + (size_t)tl * 4;
            const double w0 = hipk_peek(w4), w1 = hipk_peek(w4 + 1), w2 = hipk_peek(w4 + 2), w3 = hipk_peek(w4 + 3);
            return 0.0 + ((w0 + w1) + (w2 + w3));
        });
    };
    auto fold_subs = [&](const double *sp, int i8) {
        return hipk_fold_64x8(i8, g, [&](int ci, int ss) { return hipk_peek(sp + ci * kGmSub + ss); });
    };
    auto fold_chunks = [&](const double *cp, int i8) {   // it = 0: chunk partials of the launches before this one (hipk_reduce_parts)
        double a8 = (i8 < g) ? hipk_peek(cp + i8) : 0.0;
        a8 = 0.0 + a8;
        return hipk_wave_sum(a8);
    };
    unsigned long long seq = 0;
    int64_t it = a.it0, iters = a.it0;
    int code = 0, extra_mv = 0;
    double rs_last = scal->rs_last;
    int64_t stop_it = stop0;
    bool handed = true;   // the r / <r,r> / <rhat,r> this iteration starts from are visible (a launch or a hand-off before)
    while (it < stop_it && it - a.it0 < a.max_its) {
        if (!handed) {
            HIPK_BI_HO(a.flag_c)
            handed = true;
        }
        // ---- K1: rs = <r,r>, rho' = <rhat,r> -> tests; beta; p = r + beta (p - omega q)   (TSL:893-907)
        T rc[kBiRowRegs];
#pragma unroll
        for (int j = 0; j < kBiRowRegs; ++j)
            if (j < wmax) rc[j] = hipk_peek_off<T>(a.r, cj[j]);
        const T r_t = tlive ? hipk_peek_t<T>(a.r + trow) : (T)0;
        if (tid < 64) {
            const double v = (it == 0) ? fold_chunks(a.part_rr, tid) : fold_subs(a.part_rr, tid);
            if (tid == 0) bc[0] = v;
        } else if (tid < 128) {
            const double v = (it == 0) ? fold_chunks(a.part_rhr, tid - 64) : fold_subs(a.part_rhr, tid - 64);
            if (tid == 64) bc[1] = v;
        }
        __syncthreads();
        const double rs = bc[0], rho_new = bc[1];
        rs_last = rs;
        if (rs <= atol2) {  // TSL:894-896
            stop_it = it;
            break;
        }
        if (fabs(rho_new) < EPS * fabs(rho)) {  // TSL:902-904
            stop_it = it;
            code = -10;
            break;
        }
        const T beta = (T)(rho_new / rho * alpha / omega);  // TSL:906, left to right
        const T om = (T)omega;
        {
            const T t1 = om * q_own;
            const T t2 = p_own - t1;
            const T t3 = beta * t2;
            p_own = r_own + t3;
        }
#pragma unroll
        for (int j = 0; j < kBiRowRegs; ++j)
            if (j < wmax) {
                const T t1 = om * qc[j];
                const T t2 = pc[j] - t1;
                const T t3 = beta * t2;
                pc[j] = rc[j] + t3;
            }
        // ---- K2 (tile rows): q = A p, wavefront sums of rhat .* q   (TSL:909-910)
        T q_t;
        {
            T acc_row = (T)0;
#pragma unroll
            for (int j = 0; j < kBiRowRegs; ++j)
                if (j < wmax) {
                    const T pin = PRE ? dc[j] * pc[j] : pc[j];   // phat = M p (TSL:908)
                    const T pr = vj[j] * pin;
                    acc_row = (j < len) ? acc_row + pr : acc_row;
                }
            q_t = tlive ? acc_row : (T)0;
            double d0 = tlive ? (double)h_t * (double)q_t : 0.0;
            d0 = hipk_wave_sum(d0);
            if (lane == 0 && tile < ntiles) hipk_ho_store<LOCAL>(&a.tsum0[(size_t)tile * 4 + wave], d0);
            if (tlive) hipk_ho_store<LOCAL>(a.q + trow, q_t);
        }
        HIPK_BI_HO(a.flag_a)
        // ---- K3: alpha' = rho'/<rhat,q> -> test; s = r - alpha' q; <s,s>   (TSL:910-920)
        q_own = live ? hipk_peek_t<T>(a.q + row) : (T)0;
#pragma unroll
        for (int j = 0; j < kBiRowRegs; ++j)
            if (j < wmax) qc[j] = hipk_peek_off<T>(a.q, cj[j]);
        if (tid < 64) {
            const double v = fold_tiles(a.tsum0, tid);
            if (tid == 0) bc[2] = v;
        }
        __syncthreads();
        const double rq = bc[2];
        const double alpha_new = rho_new / rq;  // TSL:910
        if (fabs(alpha_new) < EPS) {            // TSL:913-915
            stop_it = it;
            code = -11;
            extra_mv = 1;
            break;
        }
        const T al = (T)alpha_new;
        T s_own;
        {
            const T m = al * q_own;
            s_own = r_own - m;  // TSL:917
        }
        wl[tid] = s_own;
        __syncthreads();
        if (tid < 32) {
            double acc = 0.0;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const double v = (double)wl[e * 32 + tid];
                acc = fma(v, v, acc);
            }
            acc = hipk_half_sum(acc);
            if (tid == 0) hipk_ho_store<LOCAL>(&a.part_ss[wg], acc);
        }
        // ---- K4 (tile rows): t = A s, wavefront sums of s .* t and t .* t   (TSL:923-930)
        {
            T acc_row = (T)0;
#pragma unroll
            for (int j = 0; j < kBiRowRegs; ++j)
                if (j < wmax) {
                    const T m = al * qc[j];
                    const T sc = rc[j] - m;
                    const T sin = PRE ? dc[j] * sc : sc;         // shat = M s (TSL:922)
                    const T pr = vj[j] * sin;
                    acc_row = (j < len) ? acc_row + pr : acc_row;
                }
            const T t_t = tlive ? acc_row : (T)0;
            const T mt = al * q_t;
            const T s_t = r_t - mt;
            double d0 = tlive ? (double)s_t * (double)t_t : 0.0;
            double d1 = tlive ? (double)t_t * (double)t_t : 0.0;
            d0 = hipk_wave_sum(d0);
            d1 = hipk_wave_sum(d1);
            if (lane == 0 && tile < ntiles) {
                hipk_ho_store<LOCAL>(&a.tsum2[(size_t)tile * 4 + wave], d0);
                hipk_ho_store<LOCAL>(&a.tsum1[(size_t)tile * 4 + wave], d1);
            }
            if (tlive) hipk_ho_store<LOCAL>(a.t + trow, t_t);
        }
        HIPK_BI_HO(a.flag_b)
        // ---- K5: omega' -> tests; x += alpha' p (+ omega' s); r = s (- omega' t); <r,r>, <rhat,r>   (TSL:920-961)
        const T t_own = live ? hipk_peek_t<T>(a.t + row) : (T)0;
        if (tid < 64) {
            const double v = fold_subs(a.part_ss, tid);
            if (tid == 0) bc[0] = v;
        } else if (tid < 128) {
            const double v = fold_tiles(a.tsum2, tid - 64);
            if (tid == 64) bc[1] = v;
        } else if (tid < 192) {
            const double v = fold_tiles(a.tsum1, tid - 128);
            if (tid == 128) bc[2] = v;
        }
        __syncthreads();
        const double ss = bc[0], ts = bc[1], tt = bc[2];
        const bool exit_early = ss < atol2;                                   // TSL:920 (strict)
        const double omega_new = (fabs(tt) < EPS) ? 0.0 : ts / tt;            // TSL:926-930
        if (fabs(omega_new) < EPS && !exit_early) {                           // TSL:934-936
            stop_it = it;
            code = -11;
            extra_mv = 2;
            break;
        }
        const T omn = (T)omega_new;
        const T ph_own = PRE ? d_own * p_own : p_own, sh_own = PRE ? d_own * s_own : s_own;   // x advances with phat, shat (TSL:942)
        if (exit_early) {  // TSL:942-950 with exit_early true
            const T m0 = al * ph_own;
            x_own = x_own + m0;
            r_own = s_own;
        } else {
            const T m0 = al * ph_own;
            const T m1 = omn * sh_own;
            const T m2 = m0 + m1;
            x_own = x_own + m2;
            const T m3 = omn * t_own;
            r_own = s_own - m3;
        }
        __syncthreads();   // bc and wl are free again
        wl[tid] = r_own;
        if (live) hipk_ho_store<LOCAL>(a.r + row, r_own);
        __syncthreads();
        if (tid < 32) {
            double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const double rv = (double)wl[e * 32 + tid], hv = (double)wh[e * 32 + tid];
                acc0 = fma(rv, rv, acc0);
                acc1 = fma(hv, rv, acc1);
            }
            acc0 = hipk_half_sum(acc0);
            acc1 = hipk_half_sum(acc1);
            if (tid == 0) {
                hipk_ho_store<LOCAL>(&a.part_rr[wg], acc0);
                hipk_ho_store<LOCAL>(&a.part_rhr[wg], acc1);
            }
        }
        rho = rho_new;
        alpha = alpha_new;
        omega = omega_new;
        ++it;
        iters = it;
        if (exit_early || it >= a.maxiter) {  // TSL:961, loop bound :892
            stop_it = it;
            break;
        }
        handed = false;
    }
#undef HIPK_BI_HO
    if (live) {
        a.x[row] = x_own;
        a.p[row] = p_own;
    }
    if (wg == 0 && tid == 0) {
        scal->rho = rho;
        scal->alpha = alpha;
        scal->omega = omega;
        scal->rs_last = rs_last;
        scal->iters = iters;
        scal->code = code;
        scal->extra_mv = extra_mv;
        scal->it_done = it;
        if (stop_it < stop0) scal->stop_it = stop_it;
    }
}

__global__ __launch_bounds__(HIPK_THREADS) void hipk_bi_final_kernel(hipk_bi_scal *__restrict__ scal, int g,
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

static constexpr int kBiSlots = 8;

extern "C" size_t hipk_bicgstab_work_bytes(int64_t n, int dtype) {
    const size_t sv = (dtype == HIPK_F64) ? 8 : 4;
    const size_t vec = hipk_align_up((size_t)(n > 0 ? n : 1) * sv, 256);
    // mid-size systems (hipk_bi_mid.h): q and r travel as 16-byte flagged words (q in s + t, r in two more vectors) + the partial slots
    const hipk_geom gm = hipk_make_geom(n > 0 ? n : 1);
    const bool mid = gm.g > kMidMinChunks && gm.g <= kBiMidMaxChunks;
    const size_t ll = hipk_align_up((size_t)(n > 0 ? n : 1) * 16, 256);   // q and r as 16-byte flagged words whatever the dtype
    return 256 + (size_t)kBiSlots * HIPK_MAX_PARTS * sizeof(double) + 6 * vec + (mid ? 2 * vec + 2 * ll + kBiMidSlotBytes : 0);
}
extern "C" size_t hipk_pbicgstab_work_bytes(int64_t n, int dtype) {
    const size_t sv = (dtype == HIPK_F64) ? 8 : 4;
    const size_t vec = hipk_align_up((size_t)(n > 0 ? n : 1) * sv, 256);
    return hipk_bicgstab_work_bytes(n, dtype) + 2 * vec;  // + phat, shat
}

// cb != null (PRE = false): the preconditioner is the CALLER's device code -- cb(user, in, out) enqueues out = M(in) on
// `stream` -- applied where the Jacobi variant scales in-kernel: phat = M(p) before the first SpMV, shat = M(s) before the
// second (TSL:908, 922), M(b - A x) for the final test (TSL:1007).  Same kernels, same order of operations.
template <typename T, bool PRE>
static int hipk_bicgstab_solve_t(hipk_csr_s *A, const T *dinv, const T *b, T *x, char *work, const hipk_params *prm,
                                 hipk_stats *st, hipStream_t stream, hipk_precond_fn cb = nullptr, void *user = nullptr) {
    const bool ext = cb != nullptr;
    const int64_t n = A->n_rows;
    const hipk_geom gm = A->geom;
    const size_t vec = hipk_align_up((size_t)n * sizeof(T), 256);
    hipk_bi_scal *scal = (hipk_bi_scal *)work;
    double *parts = (double *)(work + 256);
    double *part_rr = parts, *part_rhr = parts + HIPK_MAX_PARTS, *part_rq = parts + 2 * HIPK_MAX_PARTS;
    double *part_ss = parts + 3 * HIPK_MAX_PARTS, *part_ts = parts + 4 * HIPK_MAX_PARTS;
    double *part_tt = parts + 5 * HIPK_MAX_PARTS, *part_bb = parts + 6 * HIPK_MAX_PARTS;
    double *part_spare = parts + 7 * HIPK_MAX_PARTS;
    char *vbase = work + 256 + (size_t)kBiSlots * HIPK_MAX_PARTS * sizeof(double);
    T *r = (T *)vbase, *rhat = (T *)(vbase + vec), *p = (T *)(vbase + 2 * vec), *q = (T *)(vbase + 3 * vec);
    T *s = (T *)(vbase + 4 * vec), *t = (T *)(vbase + 5 * vec);
    T *phat = (PRE || ext) ? (T *)(vbase + 6 * vec) : p, *shat = (PRE || ext) ? (T *)(vbase + 7 * vec) : s;  // SpMV inputs (TSL:908, 922)

    const int64_t maxiter = (prm->maxiter < 0) ? 10 * n : prm->maxiter;
    const float tolf = (float)prm->tol, atolf = (float)prm->atol;
    const double tol2 = (double)(tolf * tolf), atol_sq = (double)(atolf * atolf);
    const int64_t check = prm->check_every > 0 ? prm->check_every : 32;

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
    int rc;

    sa.x = x;
    sa.y = r;
    sa.mode = HIPK_SPMV_RESID | HIPK_SPMV_DOT_YY;
    sa.bsub = b;
    sa.part0 = part_spare;
    sa.part1 = part_rr;
    if ((rc = hipk_launch_spmv(A, sa, stream)) != HIPK_OK) return rc;
    if ((rc = hipk_launch_dot_parts(n, b, b, A->dtype, part_bb, stream)) != HIPK_OK) return rc;
    hipk_pacer pace(A->host_poll, &scal->stop_it, check);
    HIPK_CHECK_HIP(pace.create());
    hipk_bi_start_kernel<T><<<gm.g, HIPK_THREADS, 0, stream>>>(n, gm.ch, gm.g, scal, part_rr, part_bb, part_rhr, r,
                                                                rhat, p, q, tol2, atol_sq, maxiter, pace.device_sig());
    HIPK_CHECK_HIP(hipGetLastError());

    hipk_spmv_args sq = sa, stt = sa;
    sq.x = phat;
    sq.y = q;
    sq.mode = HIPK_SPMV_DOT_W;
    sq.w = rhat;
    sq.bsub = nullptr;
    sq.part0 = part_rq;
    sq.part1 = part_spare;
    sq.stop_it = &scal->stop_it;
    stt.x = shat;
    stt.y = t;
    stt.mode = HIPK_SPMV_DOT_W | HIPK_SPMV_DOT_YY;
    stt.w = s;
    stt.bsub = nullptr;
    stt.part0 = part_ts;
    stt.part1 = part_tt;
    stt.stop_it = &scal->stop_it;

    // launch-bound systems (<= 8 reduction chunks): both SpMVs skip their combine launch, the consumers fold the tile sums
    const bool small = gm.g <= 8 && !getenv("HIPK_BICGSTAB_NO_SMALL");
    const int nt = (int)((n + HIPK_TILE - 1) / HIPK_TILE);
    const double *tsum0 = A->tile_part, *tsum1 = A->tile_part + 4 * (size_t)nt;
    sq.skip_combine = stt.skip_combine = small ? 1 : 0;

    int64_t it = 0, stop = INT64_MAX;
    // launch-bound systems of 9 .. 256 chunks (fp64, M = identity, rows of <= 12 entries within a window around their chunk): the
    // whole loop in one launch, one workgroup per chunk (hipk_bi_mid.h); HIPK_BICGSTAB_MID=0 leaves them to the paths below
    static bool mid_failed = false;
    bool mid_loop = false;
    {
        mid_loop = !ext && gm.g > kMidMinChunks && gm.g <= kBiMidMaxChunks && gm.g <= A->n_cu && gm.ch == HIPK_BASE_CHUNK && A->op_cb == nullptr &&
                   A->crow != nullptr && A->max_row_len <= 12 && prm->profile == 0 && maxiter > 0 && !mid_failed &&
                   !(getenv("HIPK_BICGSTAB_MID") && getenv("HIPK_BICGSTAB_MID")[0] == '0') && !getenv("HIPK_BICGSTAB_NO_LDS_LOOP") &&
                   !getenv("HIPK_BICGSTAB_NO_SMALL");
        void (*mid_kern)(hipk_bi_mid_args) = A->max_row_len <= 5   ? hipk_bi_mid_kernel<T, 5, PRE>
                                             : A->max_row_len <= 7 ? hipk_bi_mid_kernel<T, 7, PRE>
                                             : A->max_row_len <= 9 ? hipk_bi_mid_kernel<T, 9, PRE>
                                                                   : hipk_bi_mid_kernel<T, 12, PRE>;
        size_t lds = 0;
        hipk_mid_plan plan;
        memset(&plan, 0, sizeof(plan));
        if (mid_loop) {
            mid_loop = hipk_mid_plan_get(A, 1, stream, &plan);   // the tiles each workgroup's window holds (hipk_mid.h)
            lds = mid_loop ? hipk_bi_mid_lds_bytes(plan.max_slots * HIPK_TILE, PRE, sizeof(T)) : 0;
            int occ = 0;
            mid_loop = mid_loop && plan.max_slots <= kMidPlanSlots && lds <= (size_t)160 * 1024 &&
                       hipFuncSetAttribute((const void *)mid_kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess &&
                       hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, mid_kern, 1024, lds) == hipSuccess && (int64_t)occ * A->n_cu >= gm.g;
            (void)hipGetLastError();
        }
        if (mid_loop) {
            const char *e = getenv("HIPK_BICGSTAB_LAUNCH_ITS");
            hipk_bi_mid_args ca;
            ca.n = n;
            ca.g = gm.g;
            ca.win = plan.max_slots * HIPK_TILE;
            ca.plan = plan;
            ca.crow = A->crow;
            ca.col = A->col;
            const size_t ll_bytes = hipk_align_up((size_t)n * 16, 256);
            ca.val = A->val;
            ca.x = x;
            ca.r = r;
            ca.p = p;
            ca.q = q;
            ca.rhat = rhat;
            ca.dinv = dinv;
            ca.q_ll = (unsigned long long *)(vbase + 8 * vec);      // behind the eight vectors (hipk_bicgstab_work_bytes)
            ca.r_ll = (unsigned long long *)(vbase + 8 * vec + ll_bytes);
            ca.slots = (unsigned long long *)(vbase + 8 * vec + 2 * ll_bytes);
            ca.part_rr = part_rr;
            ca.part_rhr = part_rhr;
            ca.scal = scal;
            ca.maxiter = maxiter;
            ca.max_its = e ? atoll(e) : 8192;
            if (ca.max_its < 1) ca.max_its = 1;
            ca.slot_stride = gm.g <= 32 ? 1 : 16;
            ca.xcd_aware = 1;
            const int fail_launch = getenv("HIPK_TEST_LDS_NOT_RESIDENT") ? (atoi(getenv("HIPK_TEST_LDS_NOT_RESIDENT")) > 1 ? atoi(getenv("HIPK_TEST_LDS_NOT_RESIDENT")) : 1) : 0;
            int launch_no = 0;
            hipk_bi_scal hs0;
            for (;;) {
                ca.it0 = it;
                ca.test_not_resident = (++launch_no == fail_launch) ? 1 : 0;
                HIPK_CHECK_HIP(hipMemsetAsync(ca.q_ll, 0, 2 * ll_bytes, stream));
                HIPK_CHECK_HIP(hipMemsetAsync(ca.slots, 0, kBiMidSlotBytes, stream));
                HIPK_CHECK_HIP(hipMemsetAsync(&scal->it_done, 0, sizeof(hipk_bi_scal) - offsetof(hipk_bi_scal, it_done), stream));
                mid_kern<<<hipk_xcd_grid(gm.g), 1024, lds, stream>>>(ca);
                HIPK_CHECK_HIP(hipGetLastError());
                HIPK_CHECK_HIP(hipMemcpyAsync(&hs0, scal, sizeof(hs0), hipMemcpyDeviceToHost, stream));
                HIPK_CHECK_HIP(hipStreamSynchronize(stream));
                if (hs0.redo < 0) {
                    if (hs0.redo == -3) {
                        hipk_set_error("hipk_bicgstab_solve: a resident workgroup of the one-launch loop stopped arriving");
                        return HIPK_ERR_HIP;
                    }
                    if (!getenv("HIPK_TEST_LDS_NOT_RESIDENT")) mid_failed = true;   // not co-resident; this launch modified nothing
                    mid_loop = false;
                    break;
                }
                it = hs0.it_done;
                if (hs0.stop_it <= it || it >= maxiter) break;
            }
        }
    }
    // launch-bound systems with short rows, M = identity: the whole loop in one launch (hipk_bi_solve_lds_kernel)
    static bool lds_loop_failed = false;   // its workgroups once failed to meet (a shared device): do not wait for that verdict again
    // up to 64 workgroups (8 chunks) on ONE XCD; up to 32 chunks (n <= 65536) spread over the chip, two workgroups per compute unit
    const bool lds_spread = kGmSub * gm.g > 64;
    bool lds_loop = gm.g <= 32 && !getenv("HIPK_BICGSTAB_NO_SMALL") && !ext && gm.ch == HIPK_BASE_CHUNK &&
                    A->max_row_len <= kBiRowRegs && prm->profile == 0 && maxiter > 0 &&
                    kGmSub * gm.g <= (lds_spread ? 2 * A->n_cu : 2 * (A->n_cu / 8)) && !lds_loop_failed &&
                    !getenv("HIPK_BICGSTAB_NO_LDS_LOOP") && !(lds_spread && getenv("HIPK_NO_LDS_SPREAD")) && !mid_loop &&
                    it == 0;   // (after a one-launch loop above gave up mid-solve, part_rr / part_rhr hold CHUNK partials: launch sequence)
    if (lds_loop) {
        bool local = !lds_spread && !getenv("HIPK_BICGSTAB_LOOP_AGENT");
        const char *e = getenv("HIPK_BICGSTAB_LAUNCH_ITS");
        hipk_bi_lds_args<T> ca;
        ca.n = n;
        ca.g = gm.g;
        ca.crow = A->crow;
        ca.col = A->col;
        ca.val = (const T *)A->val;
        ca.x = x;
        ca.r = r;
        ca.p = p;
        ca.q = q;
        ca.t = t;
        ca.rhat = rhat;
        ca.dinv = dinv;
        ca.scal = scal;
        ca.tsum0 = A->tile_part;
        ca.tsum1 = A->tile_part + 4 * (size_t)nt;
        ca.tsum2 = part_ts;   // (the launch sequence's chunk-partial slots of <t,s>, <t,t> are free here)
        ca.part_rr = part_rr;
        ca.part_rhr = part_rhr;
        ca.part_ss = part_ss;
        ca.flag_a = (unsigned long long *)part_tt;   // 3 x 512 words
        ca.flag_b = ca.flag_a + kHoMaxWg;
        ca.flag_c = ca.flag_a + 2 * kHoMaxWg;
        ca.spread = lds_spread ? 1 : 0;
        const int lgrid = lds_spread ? kGmSub * gm.g : 8 * kGmSub * gm.g;
        ca.maxiter = maxiter;
        ca.max_its = e ? atoll(e) : 8192;
        if (ca.max_its < 1) ca.max_its = 1;
        // tests: HIPK_TEST_LDS_NOT_RESIDENT=k makes the k-th launch of this solve report its workgroups as not co-resident
        const int fail_launch = getenv("HIPK_TEST_LDS_NOT_RESIDENT") ? (atoi(getenv("HIPK_TEST_LDS_NOT_RESIDENT")) > 1 ? atoi(getenv("HIPK_TEST_LDS_NOT_RESIDENT")) : 1) : 0;
        int launch_no = 0;
        hipk_bi_scal hs0;
        for (;;) {
            ca.it0 = it;
            ca.test_not_resident = (++launch_no == fail_launch) ? 1 : 0;
            HIPK_CHECK_HIP(hipMemsetAsync(ca.flag_a, 0, 3 * kHoMaxWg * sizeof(unsigned long long), stream));
            HIPK_CHECK_HIP(hipMemsetAsync(&scal->it_done, 0, sizeof(hipk_bi_scal) - offsetof(hipk_bi_scal, it_done), stream));
            if (local)
                hipk_bi_solve_lds_kernel<T, true, PRE><<<lgrid, HIPK_THREADS, 0, stream>>>(ca);
            else
                hipk_bi_solve_lds_kernel<T, false, PRE><<<lgrid, HIPK_THREADS, 0, stream>>>(ca);
            HIPK_CHECK_HIP(hipGetLastError());
            HIPK_CHECK_HIP(hipMemcpyAsync(&hs0, scal, sizeof(hs0), hipMemcpyDeviceToHost, stream));
            HIPK_CHECK_HIP(hipStreamSynchronize(stream));
            if (hs0.redo < 0) {
                if (hs0.redo == -3) {
                    hipk_set_error("hipk_bicgstab_solve: a resident workgroup of the one-launch loop stopped arriving");
                    return HIPK_ERR_HIP;
                }
                if (hs0.redo == -2 && local) {   // spread over several XCDs: agent-scope hand-offs
                    local = false;
                    continue;
                }
                if (!getenv("HIPK_TEST_LDS_NOT_RESIDENT")) lds_loop_failed = true;   // not co-resident; this launch modified nothing: the launch sequence below takes over
                if (it > 0) {
                    // ... from iteration `it` of an EARLIER launch: the vectors and scalars are in memory, but part_rr / part_rhr hold
                    // that launch's 8 g SUB-partials, not the g chunk partials the direction kernel folds.  Recompute them from r
                    // and rhat (the spec's plain dot: the bits the x-update kernel of the launch sequence would have left)
                    if ((rc = hipk_launch_dot_parts(n, r, r, A->dtype, part_rr, stream)) != HIPK_OK) return rc;
                    if ((rc = hipk_launch_dot_parts(n, rhat, r, A->dtype, part_rhr, stream)) != HIPK_OK) return rc;
                }
                lds_loop = false;
                break;
            }
            it = hs0.it_done;
            if (hs0.stop_it <= it || it >= maxiter) break;
        }
    }
    if (mid_loop) lds_loop = true;   // finished in the one-launch loop
    for (; !lds_loop && it < maxiter; ++it) {
        HIPK_CHECK_HIP(pace.gate(it, stream, &stop));
        if (stop <= it) break;
        {
            hipk_bi_direction_kernel<T, PRE><<<gm.g, HIPK_THREADS, 0, stream>>>(n, gm.ch, gm.g, scal, it, part_rr,
                                                                                 part_rhr, r, q, p, dinv, phat);
            if (ext && cb(user, p, phat) != 0) {
                hipk_set_error("hipk_pbicgstab_solve_cb: the preconditioner callback failed");
                return HIPK_ERR_ARG;
            }
            sq.it = it;
            if ((rc = hipk_launch_spmv(A, sq, stream, &prof)) != HIPK_OK) return rc;
            if (small)
                hipk_bi_supdate_kernel<T, PRE, true><<<gm.g, HIPK_THREADS, 0, stream>>>(n, gm.ch, gm.g, scal, it, part_rhr, tsum0, r,
                                                                                        q, s, part_ss, dinv, shat, nt);
            else
                hipk_bi_supdate_kernel<T, PRE, false><<<gm.g, HIPK_THREADS, 0, stream>>>(n, gm.ch, gm.g, scal, it, part_rhr, part_rq,
                                                                                         r, q, s, part_ss, dinv, shat, 0);
            if (ext && cb(user, s, shat) != 0) {
                hipk_set_error("hipk_pbicgstab_solve_cb: the preconditioner callback failed");
                return HIPK_ERR_ARG;
            }
            stt.it = it;
            if ((rc = hipk_launch_spmv(A, stt, stream)) != HIPK_OK) return rc;
            // x advances with phat / shat (TSL:942): the PRE form of the kernel also when they come from the callback
            if (small && (PRE || ext))
                hipk_bi_xupdate_kernel<T, true, true><<<gm.g, HIPK_THREADS, 0, stream>>>(
                    n, gm.ch, gm.g, scal, it, maxiter, part_ss, tsum0, tsum1, phat, s, t, rhat, x, r, part_rr, part_rhr, shat, nt);
            else if (small)
                hipk_bi_xupdate_kernel<T, false, true><<<gm.g, HIPK_THREADS, 0, stream>>>(
                    n, gm.ch, gm.g, scal, it, maxiter, part_ss, tsum0, tsum1, phat, s, t, rhat, x, r, part_rr, part_rhr, shat, nt);
            else if (PRE || ext)
                hipk_bi_xupdate_kernel<T, true, false><<<gm.g, HIPK_THREADS, 0, stream>>>(
                    n, gm.ch, gm.g, scal, it, maxiter, part_ss, part_ts, part_tt, phat, s, t, rhat, x, r, part_rr, part_rhr, shat, 0);
            else
                hipk_bi_xupdate_kernel<T, false, false><<<gm.g, HIPK_THREADS, 0, stream>>>(
                    n, gm.ch, gm.g, scal, it, maxiter, part_ss, part_ts, part_tt, phat, s, t, rhat, x, r, part_rr, part_rhr, shat, 0);
        }
        if ((it & 31) == 31) HIPK_CHECK_HIP(hipGetLastError());
    }
    HIPK_CHECK_HIP(hipGetLastError());

    // TSL:1007-1014 (PRE: ||M (b - A x)||, the row scaling runs in the SpMV epilogue)
    sa.x = x;
    sa.y = t;
    sa.mode = HIPK_SPMV_RESID | HIPK_SPMV_DOT_YY | (PRE ? HIPK_SPMV_SCALE : 0);
    sa.dscale = dinv;
    sa.bsub = b;
    sa.part0 = part_spare;
    sa.part1 = part_ss;
    sa.stop_it = nullptr;
    if ((rc = hipk_launch_spmv(A, sa, stream)) != HIPK_OK) return rc;
    if (ext) {  // ||M (b - A x)||^2 of the caller's M
        if (cb(user, t, phat) != 0) {
            hipk_set_error("hipk_pbicgstab_solve_cb: the preconditioner callback failed");
            return HIPK_ERR_ARG;
        }
        if ((rc = hipk_launch_dot_parts(n, phat, phat, A->dtype, part_ss, stream)) != HIPK_OK) return rc;
    }
    if ((rc = hipk_launch_dot_parts(n, x, x, A->dtype, part_bb, stream)) != HIPK_OK) return rc;
    hipk_bi_final_kernel<<<1, HIPK_THREADS, 0, stream>>>(scal, gm.g, part_ss, part_bb);
    HIPK_CHECK_HIP(hipGetLastError());
    hipk_bi_scal hs;
    HIPK_CHECK_HIP(hipEventRecord(whole.b, stream));
    HIPK_CHECK_HIP(hipMemcpyAsync(&hs, scal, sizeof(hs), hipMemcpyDeviceToHost, stream));
    HIPK_CHECK_HIP(hipStreamSynchronize(stream));

    hipk_finish_isolve_stats(st, prm, hs.bs, hs.res2, hs.xx, hs.iters, 1 + 2 * hs.iters + hs.extra_mv + 1);
    st->recurrence_rs = hs.rs_last;
    st->breakdown = hs.code;
    float ms = 0.f;
    HIPK_CHECK_HIP(hipEventElapsedTime(&ms, whole.a, whole.b));
    st->solve_ms = ms;
    HIPK_CHECK_HIP(prof.collect(st, hs.iters));
    return HIPK_OK;
}

extern "C" int hipk_bicgstab_solve(hipk_csr_t A, const void *b, void *x, void *work, size_t work_bytes,
                                   const hipk_params *prm, hipk_stats *st, hipk_stream_t stream) {
    HIPK_REQUIRE(A && b && x && work && prm && st, HIPK_ERR_ARG, "null argument");
    HIPK_REQUIRE(A->n_rows == A->n_cols, HIPK_ERR_ARG, "linear operator must be a square matrix");
    HIPK_REQUIRE(A->n_rows > 0, HIPK_ERR_ARG, "empty system");
    HIPK_REQUIRE(hipk_aligned16(b) && hipk_aligned16(x) && (((uintptr_t)work) & 255u) == 0, HIPK_ERR_ALIGN,
                 "b/x must be 16-byte and work 256-byte aligned");
    HIPK_REQUIRE(work_bytes >= hipk_bicgstab_work_bytes(A->n_rows, A->dtype), HIPK_ERR_WORKSPACE, "work too small");
    HIPK_REQUIRE(b != x, HIPK_ERR_ARG, "b and x must not alias");
    memset(st, 0, sizeof(*st));
    if (A->dtype == HIPK_F64)
        return hipk_bicgstab_solve_t<double, false>(A, nullptr, (const double *)b, (double *)x, (char *)work, prm, st,
                                                    (hipStream_t)stream);
    return hipk_bicgstab_solve_t<float, false>(A, nullptr, (const float *)b, (float *)x, (char *)work, prm, st,
                                               (hipStream_t)stream);
}

extern "C" int hipk_pbicgstab_solve(hipk_csr_t A, const void *dinv, const void *b, void *x, void *work, size_t work_bytes,
                                    const hipk_params *prm, hipk_stats *st, hipk_stream_t stream) {
    HIPK_REQUIRE(A && dinv && b && x && work && prm && st, HIPK_ERR_ARG, "null argument");
    HIPK_REQUIRE(A->n_rows == A->n_cols, HIPK_ERR_ARG, "linear operator must be a square matrix");
    HIPK_REQUIRE(A->n_rows > 0, HIPK_ERR_ARG, "empty system");
    HIPK_REQUIRE(hipk_aligned16(b) && hipk_aligned16(x) && hipk_aligned16(dinv) && (((uintptr_t)work) & 255u) == 0,
                 HIPK_ERR_ALIGN, "b/x/dinv must be 16-byte and work 256-byte aligned");
    HIPK_REQUIRE(work_bytes >= hipk_pbicgstab_work_bytes(A->n_rows, A->dtype), HIPK_ERR_WORKSPACE, "work too small");
    HIPK_REQUIRE(b != x, HIPK_ERR_ARG, "b and x must not alias");
    memset(st, 0, sizeof(*st));
    if (A->dtype == HIPK_F64)
        return hipk_bicgstab_solve_t<double, true>(A, (const double *)dinv, (const double *)b, (double *)x, (char *)work, prm,
                                                   st, (hipStream_t)stream);
    return hipk_bicgstab_solve_t<float, true>(A, (const float *)dinv, (const float *)b, (float *)x, (char *)work, prm, st,
                                              (hipStream_t)stream);
}

extern "C" int hipk_pbicgstab_solve_cb(hipk_csr_t A, hipk_precond_fn M, void *user, const void *b, void *x, void *work,
                                       size_t work_bytes, const hipk_params *prm, hipk_stats *st, hipk_stream_t stream) {
    HIPK_REQUIRE(A && M && b && x && work && prm && st, HIPK_ERR_ARG, "null argument");
    HIPK_REQUIRE(A->n_rows == A->n_cols, HIPK_ERR_ARG, "linear operator must be a square matrix");
    HIPK_REQUIRE(A->n_rows > 0, HIPK_ERR_ARG, "empty system");
    HIPK_REQUIRE(hipk_aligned16(b) && hipk_aligned16(x) && (((uintptr_t)work) & 255u) == 0, HIPK_ERR_ALIGN,
                 "b/x must be 16-byte and work 256-byte aligned");
    HIPK_REQUIRE(work_bytes >= hipk_pbicgstab_work_bytes(A->n_rows, A->dtype), HIPK_ERR_WORKSPACE, "work too small");
    HIPK_REQUIRE(b != x, HIPK_ERR_ARG, "b and x must not alias");
    memset(st, 0, sizeof(*st));
    if (A->dtype == HIPK_F64)
        return hipk_bicgstab_solve_t<double, false>(A, nullptr, (const double *)b, (double *)x, (char *)work, prm, st,
                                                    (hipStream_t)stream, M, user);
    return hipk_bicgstab_solve_t<float, false>(A, nullptr, (const float *)b, (float *)x, (char *)work, prm, st,
                                               (hipStream_t)stream, M, user);
}


// =====================================================================================================================
// Row-partitioned BiCGStab, the loop of one rank in C (new against the reference, which is single-device; SURVEY 8e names CG,
// VERDICT r1 lists this as the next solver to shard).  The algorithm is `bicgstab` (TSL:859-964 via `_isolve`, TSL:968-1016) on the
// SAME five kernels as the single-GPU solve; every kernel folds the chunk partials of ALL ranks (gathered in global chunk order),
// so the iterates are bitwise those of the single-GPU solve for any rank count.  Per iteration, on the solver's stream:
//   all-gather <r,r>, <rhat,r> (one group) | K1 | halo of p | SpMV q, all-gather <rhat,q> | K3 | ONE group: all-gather <s,s> + halo
//   of s | SpMV t, all-gather <t,s>, <t,t> (one group) | K5
// -- five collective launches (CG: two).  Conventions of hipk_dist_cg_solve (csrc/hipk_dist.hip): fixed batches, the stop word
// read one batch late, identical decisions on all ranks without an agreement collective.
#include <vector>

struct hipk_dbi_layout {
    size_t scal, part_a, part_b, spare, g_rr, g_rhr, g_rq, g_ss, g_ts, g_tt, g_bb, out4, send_buf, slab_loc, slab_all;
    size_t r, rhat, p, q, s, t, total;
};
static hipk_dbi_layout hipk_dbi_make_layout(const hipk_dist_plan *pl) {
    hipk_dbi_layout L;
    size_t o = 0;
    auto take = [&](size_t bytes) {
        const size_t at = o;
        o += hipk_align_up(bytes, 256);
        return at;
    };
    const size_t per = (size_t)pl->per, W = (size_t)pl->world;
    const size_t next = (size_t)(pl->n_ext > 0 ? pl->n_ext : 1), nloc = (size_t)(pl->n_local > 0 ? pl->n_local : 1);
    L.scal = take(256);
    L.part_a = take(per * 8);
    L.part_b = take(per * 8);
    L.spare = take(per * 8);
    L.g_rr = take(W * per * 8);
    L.g_rhr = take(W * per * 8);
    L.g_rq = take(W * per * 8);
    L.g_ss = take(W * per * 8);
    L.g_ts = take(W * per * 8);
    L.g_tt = take(W * per * 8);
    L.g_bb = take(W * per * 8);
    L.out4 = take(4 * 8);
    L.send_buf = take((size_t)(pl->n_send > 0 ? pl->n_send : 1) * 8);
    L.slab_loc = take((size_t)(pl->slab > 0 ? pl->slab : 1) * 8);
    L.slab_all = take((size_t)(pl->slab > 0 ? pl->slab : 1) * W * 8);
    L.r = take(nloc * 8);
    L.rhat = take(nloc * 8);
    L.p = take(next * 8);
    L.q = take(nloc * 8);
    L.s = take(next * 8);
    L.t = take(next * 8);   // also the x-halo scratch of the residual SpMVs
    L.total = o;
    return L;
}
extern "C" size_t hipk_dist_bicgstab_work_bytes(const hipk_dist_plan *plan) {
    if (!plan || plan->world < 1 || plan->per < 1) return 0;
    return hipk_dbi_make_layout(plan).total;
}

#define HIPK_DBI_NCCL(expr, what)                                                          \
    do {                                                                                   \
        const int _r = (expr);                                                             \
        if (_r != 0) {                                                                     \
            hipk_set_error("hipk_dist_bicgstab_solve: %s failed (ncclResult %d)", what, _r); \
            return HIPK_ERR_HIP;                                                           \
        }                                                                                  \
    } while (0)
#define HIPK_DBI_TRY(expr)              \
    do {                                \
        const int _rc = (expr);         \
        if (_rc != HIPK_OK) return _rc; \
    } while (0)

extern "C" int hipk_dist_bicgstab_solve(hipk_csr_t A, const hipk_dist_plan *pl, const hipk_rccl *cc, const void *b_local, void *x_ext,
                                        void *work, size_t work_bytes, const hipk_params *prm, hipk_stats *st, hipk_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    HIPK_REQUIRE(A && pl && cc && b_local && x_ext && work && prm && st, HIPK_ERR_ARG, "null argument");
    HIPK_REQUIRE(A->dtype == HIPK_F64, HIPK_ERR_UNSUPPORTED, "the row-partitioned solver is fp64");
    HIPK_REQUIRE(pl->world >= 1 && pl->rank >= 0 && pl->rank < pl->world, HIPK_ERR_ARG, "rank / world");
    HIPK_REQUIRE(pl->n_local > 0 && pl->n_local == A->n_rows && pl->n_ext >= pl->n_local, HIPK_ERR_ARG,
                 "every rank must own rows (n_local > 0) and n_ext >= n_local");
    HIPK_REQUIRE(pl->per >= 1 && (int64_t)pl->per * pl->world >= pl->g_red && pl->g_red >= 1 && pl->g_red <= HIPK_MAX_PARTS,
                 HIPK_ERR_ARG, "partial-sum geometry");
    HIPK_REQUIRE((pl->n_local + pl->chunk_rows - 1) / pl->chunk_rows <= pl->per, HIPK_ERR_ARG, "more local chunks than `per`");
    HIPK_REQUIRE(cc->all_gather && cc->group_start && cc->group_end && (pl->world == 1 || pl->halo_mode == 0 || (cc->send && cc->recv)),
                 HIPK_ERR_ARG, "missing collective entry points");
    HIPK_REQUIRE((((uintptr_t)work) & 255u) == 0 && hipk_aligned16(x_ext) && hipk_aligned16(b_local), HIPK_ERR_ALIGN,
                 "work must be 256-byte, x / b 16-byte aligned");
    const hipk_dbi_layout L = hipk_dbi_make_layout(pl);
    HIPK_REQUIRE(work_bytes >= L.total, HIPK_ERR_WORKSPACE, "work too small");
    memset(st, 0, sizeof(*st));
    typedef double T;
    char *wk = (char *)work;
    hipk_bi_scal *scal = (hipk_bi_scal *)(wk + L.scal);
    double *part_a = (double *)(wk + L.part_a), *part_b = (double *)(wk + L.part_b), *spare = (double *)(wk + L.spare);
    double *g_rr = (double *)(wk + L.g_rr), *g_rhr = (double *)(wk + L.g_rhr), *g_rq = (double *)(wk + L.g_rq);
    double *g_ss = (double *)(wk + L.g_ss), *g_ts = (double *)(wk + L.g_ts), *g_tt = (double *)(wk + L.g_tt);
    double *g_bb = (double *)(wk + L.g_bb), *out4 = (double *)(wk + L.out4);
    double *send_buf = (double *)(wk + L.send_buf), *slab_loc = (double *)(wk + L.slab_loc), *slab_all = (double *)(wk + L.slab_all);
    T *r = (T *)(wk + L.r), *rhat = (T *)(wk + L.rhat), *p = (T *)(wk + L.p), *q = (T *)(wk + L.q), *s = (T *)(wk + L.s);
    T *t = (T *)(wk + L.t);
    T *x = (T *)x_ext;
    const T *b = (const T *)b_local;
    const int64_t n = pl->n_local, n_ext = pl->n_ext;
    const int ch = pl->chunk_rows, G = pl->g_red, per = pl->per, W = pl->world;
    const int grid = (int)((n + ch - 1) / ch);
    const int64_t maxiter = (prm->maxiter < 0) ? 10 * pl->n_global : prm->maxiter;   // TSL:982-984
    const float tolf = (float)prm->tol, atolf = (float)prm->atol;
    const double tol2 = (double)(tolf * tolf), atol_sq = (double)(atolf * atolf);
    const int NCCL_F64 = 8;
    enum { MODE_DOT_W = 1, MODE_DOT_YY = 2, MODE_RESID = 4 };

    hipk_event_pair whole;
    HIPK_CHECK_HIP(whole.create());
    HIPK_CHECK_HIP(hipEventRecord(whole.a, stream));
    HIPK_CHECK_HIP(hipMemsetAsync(wk, 0, L.total, stream));
    if (n_ext > n) HIPK_CHECK_HIP(hipMemsetAsync(x + n, 0, (size_t)(n_ext - n) * 8, stream));

    auto gather = [&](const double *src, double *dst) -> int {
        HIPK_DBI_NCCL(cc->all_gather(src, dst, (size_t)per, NCCL_F64, cc->comm, stream), "all_gather(partials)");
        return HIPK_OK;
    };
    bool need_pack = false;
    for (int peer = 0; peer < W; ++peer)
        if (pl->send_counts[peer] > 0 && !(pl->send_first && pl->send_first[peer] >= 0)) need_pack = true;
    auto halo_p2p_calls = [&](double *v) -> int {
        size_t so = 0, ro = 0;
        for (int peer = 0; peer < W; ++peer) {
            const size_t ns = (size_t)pl->send_counts[peer], nr = (size_t)pl->recv_counts[peer];
            const bool direct = pl->send_first && pl->send_first[peer] >= 0;
            if (ns) HIPK_DBI_NCCL(cc->send(direct ? v + pl->send_first[peer] : send_buf + so, ns, NCCL_F64, peer, cc->comm, stream), "send(halo)");
            if (nr) HIPK_DBI_NCCL(cc->recv(v + n + ro, nr, NCCL_F64, peer, cc->comm, stream), "recv(halo)");
            so += ns;
            ro += nr;
        }
        return HIPK_OK;
    };
    // the peers' entries this rank's rows reference -> v[n .. n_ext), optionally with all-gathers of partials in the same group
    auto exchange = [&](double *v, const double *ps0, double *pd0) -> int {
        if (W == 1) {
            if (ps0) HIPK_DBI_TRY(gather(ps0, pd0));
            return HIPK_OK;
        }
        const bool halo = v != nullptr && !(pl->n_send == 0 && pl->n_ghost == 0 && pl->halo_mode == 1);
        if (halo && pl->halo_mode == 1 && pl->n_send && need_pack)
            HIPK_DBI_TRY(hipk_gather(pl->n_send, pl->send_idx_dev, v, send_buf, HIPK_F64, stream));
        if (halo && pl->halo_mode == 0 && pl->n_send)
            HIPK_DBI_TRY(hipk_gather(pl->n_send, pl->send_idx_dev, v, slab_loc, HIPK_F64, stream));
        HIPK_DBI_NCCL(cc->group_start(), "group_start");
        if (ps0) HIPK_DBI_NCCL(cc->all_gather(ps0, pd0, (size_t)per, NCCL_F64, cc->comm, stream), "all_gather(partials)");
        if (halo && pl->halo_mode == 1) HIPK_DBI_TRY(halo_p2p_calls(v));
        if (halo && pl->halo_mode == 0)
            HIPK_DBI_NCCL(cc->all_gather(slab_loc, slab_all, (size_t)pl->slab, NCCL_F64, cc->comm, stream), "all_gather(halo slabs)");
        HIPK_DBI_NCCL(cc->group_end(), "group_end");
        if (halo && pl->halo_mode == 0 && pl->n_ghost)
            HIPK_DBI_TRY(hipk_gather(pl->n_ghost, pl->ghost_src_dev, slab_all, v + n, HIPK_F64, stream));
        return HIPK_OK;
    };
    auto gather2 = [&](const double *a0, double *d0, const double *a1, double *d1) -> int {
        if (W > 1) HIPK_DBI_NCCL(cc->group_start(), "group_start");
        HIPK_DBI_NCCL(cc->all_gather(a0, d0, (size_t)per, NCCL_F64, cc->comm, stream), "all_gather(partials)");
        HIPK_DBI_NCCL(cc->all_gather(a1, d1, (size_t)per, NCCL_F64, cc->comm, stream), "all_gather(partials)");
        if (W > 1) HIPK_DBI_NCCL(cc->group_end(), "group_end");
        return HIPK_OK;
    };

    // ---- r0 = b - A x0 with <r0,r0>; <b,b>; rhat = p = q = r0, <rhat,r0> = <r0,r0>   (TSL:870-890)
    // (x carries its halo tail for the residual forms; t doubles as nothing here: x_ext is the caller's n_ext vector)
    HIPK_DBI_TRY(exchange(x, nullptr, nullptr));
    HIPK_DBI_TRY(hipk_spmv_ex(A, x, r, MODE_RESID | MODE_DOT_YY, nullptr, b, spare, part_a, nullptr, 0, stream));
    HIPK_DBI_TRY(gather(part_a, g_rr));
    HIPK_DBI_TRY(hipk_dot_parts(n, ch, b, b, HIPK_F64, part_a, stream));
    HIPK_DBI_TRY(gather(part_a, g_bb));
    HIPK_CHECK_HIP(hipMemcpyAsync(g_rhr, g_rr, (size_t)W * per * 8, hipMemcpyDeviceToDevice, stream));
    hipk_bi_start_kernel<T><<<grid, HIPK_THREADS, 0, stream>>>(n, ch, G, scal, g_rr, g_bb, spare, r, rhat, p, q, tol2, atol_sq, maxiter,
                                                                nullptr);
    HIPK_CHECK_HIP(hipGetLastError());

    // ---- the loop: fixed batches, the stop word read one batch late (two reads in flight)
    int64_t batch = prm->check_every > 0 ? prm->check_every : 16;
    hipk_poller poll(A->host_poll);
    HIPK_CHECK_HIP(poll.create());
    const int64_t *stop_dev = &scal->stop_it;
    int64_t it = 0, stop = INT64_MAX;
    while (it < maxiter) {
        const int64_t end = (it + batch < maxiter) ? it + batch : maxiter;
        for (; it < end; ++it) {
            hipk_bi_direction_kernel<T, false><<<grid, HIPK_THREADS, 0, stream>>>(n, ch, G, scal, it, g_rr, g_rhr, r, q, p, nullptr, p);
            HIPK_DBI_TRY(exchange(p, nullptr, nullptr));
            HIPK_DBI_TRY(hipk_spmv_ex(A, p, q, MODE_DOT_W, rhat, nullptr, part_a, spare, stop_dev, it, stream));
            HIPK_DBI_TRY(gather(part_a, g_rq));
            hipk_bi_supdate_kernel<T, false, false><<<grid, HIPK_THREADS, 0, stream>>>(n, ch, G, scal, it, g_rhr, g_rq, r, q, s, part_a,
                                                                                      nullptr, s, 0);
            HIPK_DBI_TRY(exchange(s, part_a, g_ss));
            HIPK_DBI_TRY(hipk_spmv_ex(A, s, t, MODE_DOT_W | MODE_DOT_YY, s, nullptr, part_a, part_b, stop_dev, it, stream));
            HIPK_DBI_TRY(gather2(part_a, g_ts, part_b, g_tt));
            hipk_bi_xupdate_kernel<T, false, false><<<grid, HIPK_THREADS, 0, stream>>>(n, ch, G, scal, it, maxiter, g_ss, g_ts, g_tt, p, s, t,
                                                                                      rhat, x, r, part_a, part_b, s, 0);
            HIPK_DBI_TRY(gather2(part_a, g_rr, part_b, g_rhr));
        }
        HIPK_CHECK_HIP(hipGetLastError());
        HIPK_CHECK_HIP(poll.post(stop_dev, it, stream));
        if (poll.count == 2) {
            HIPK_CHECK_HIP(hipEventSynchronize(poll.ev[poll.head]));
            poll.harvest(&stop);
        }
        if (stop <= it - batch) break;
    }
    HIPK_CHECK_HIP(poll.drain(&stop));

    // ---- TSL:1007-1014: true residual and ||x|| decide info
    HIPK_DBI_TRY(exchange(x, nullptr, nullptr));
    HIPK_DBI_TRY(hipk_spmv_ex(A, x, t, MODE_RESID | MODE_DOT_YY, nullptr, b, spare, part_a, nullptr, 0, stream));
    HIPK_DBI_TRY(gather(part_a, g_ss));
    HIPK_DBI_TRY(hipk_reduce_parts(g_ss, G, out4 + 0, stream));
    HIPK_DBI_TRY(hipk_dot_parts(n, ch, x, x, HIPK_F64, part_a, stream));
    HIPK_DBI_TRY(gather(part_a, g_ts));
    HIPK_DBI_TRY(hipk_reduce_parts(g_ts, G, out4 + 1, stream));
    HIPK_DBI_TRY(hipk_reduce_parts(g_bb, G, out4 + 2, stream));
    double h4[4] = {0, 0, 0, 0};
    hipk_bi_scal hs;
    HIPK_CHECK_HIP(hipEventRecord(whole.b, stream));
    HIPK_CHECK_HIP(hipMemcpyAsync(h4, out4, sizeof(h4), hipMemcpyDeviceToHost, stream));
    HIPK_CHECK_HIP(hipMemcpyAsync(&hs, scal, sizeof(hs), hipMemcpyDeviceToHost, stream));
    HIPK_CHECK_HIP(hipStreamSynchronize(stream));
    hipk_finish_isolve_stats(st, prm, h4[2], h4[0], h4[1], hs.iters, 1 + 2 * hs.iters + hs.extra_mv + 1);
    st->recurrence_rs = hs.rs_last;
    st->breakdown = hs.code;
    float ms = 0.f;
    HIPK_CHECK_HIP(hipEventElapsedTime(&ms, whole.a, whole.b));
    st->solve_ms = ms;
    return HIPK_OK;
}
