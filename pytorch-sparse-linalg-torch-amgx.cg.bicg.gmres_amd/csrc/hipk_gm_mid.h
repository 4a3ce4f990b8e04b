// hipk_gm_mid.h -- GMRES for launch-bound MID-SIZE systems (9 .. 256 reduction chunks, fp64, no preconditioner, restart <= 31,
// banded rows of <= 12 entries): THE ARNOLDI STEPS OF A RESTART CYCLE IN ONE LAUNCH, one 1024-thread workgroup per reduction
// chunk -- the scheme of hipk_cg_mid.h (flagged 16-byte words, an LDS window of the vector the product gathers from, every
// workgroup folds all chunk partials itself) applied to the step loop of hipk_gmres.hip:
//   per step k:  w = A v_k (own rows; v_k is the LDS window), tile sums of w .* w                         -> partial of ||A v||^2
//     CGS pass:  h_j = <V_j, w> chains of the own chunk, j <= k (the basis stays in memory: a workgroup only ever re-reads ITS
//                rows of it, plain cached loads)                          -> hand-off: k + 1 chunk partials, folded per column
//                q = w - V h (own rows), <q,q> chains                     -> hand-off: the partial of <q,q> (+ ||A v||^2)
//                second pass iff ||r|| < ||q|| / sqrt 2 (TSL:313-326), decided by every workgroup from the same bits
//     normalise: v_{k+1} = q / ||q|| (own rows: to the basis, to the window, to the neighbours as flagged words), column k of H,
//                Givens, breakdown and early-exit tests on every workgroup's private copy of the small arrays (workgroup 0
//                also keeps the solve's header block up to date: the host finishes the cycle from it as after the launches)
//                                                                          -> hand-off: v_{k+1} at the window's halo columns
// Three hand-offs per step (five with a second pass) instead of five to nine launches.  Arithmetic per element and every fold as
// in hipk_gm_multidot_stream_kernel / hipk_gm_hreduce_kernel / hipk_gm_update_stream_kernel / hipk_gm_decide_kernel /
// hipk_gm_normalize_kernel, bit for bit; tests/test_gpu_api.py::test_gmres_mid_one_launch_cycle_is_bit_identical.
#ifndef HIPK_GM_MID_H
#define HIPK_GM_MID_H
#include "hipk_mid.h"

static constexpr int kGmMidMinChunks = 32;    // up to 32 chunks the whole-solve kernel (hipk_gm_solve_lds_kernel, spread) is faster: GMRES(30) ms per
                                              // cycle, that kernel / this one: 0.49 / 0.61 at 11 chunks, 0.53 / 0.59 at 20, 0.59 / 0.60 at 31
                                              // (HIPK_GMRES_MID_MIN=8 forces this one; profiles/r03_gmres_mid_vs_whole_solve_9_32_chunks.jsonl)
static constexpr int kGmMidMaxChunks = 256;   // one chunk per workgroup, one workgroup per CU
static constexpr int kGmMidCols = 32;         // restart <= 31: columns 0 .. 31 of the basis
// slot arrays: <V_j, w> [32][g] | <q,q> [g] | <w,w> [2][g], each slot up to 256 bytes
static constexpr int kGmMidKinds = kGmMidCols + 3;   // + <q,q> + ||A v||^2 twice (by step parity: it is consumed a hand-off after it is published)
static constexpr size_t kGmMidSlotBytes = (size_t)kGmMidKinds * kGmMidMaxChunks * 256;

struct hipk_gm_mid_args {
    int64_t n;
    int g, win, m;                 // chunks; doubles of the window in LDS (256 x the most tiles any workgroup's window holds); restart
    hipk_mid_plan plan;            // which tiles (hipk_mid.h)
    const int *crow, *col;
    const void *val;               // values, basis, dinv: of the handle's dtype (the kernel's T)
    void *V;                       // the basis, column j at V + j ldv
    int64_t ldv;
    unsigned long long *v_ll;      // [2 n] flagged words of v_{k+1}
    unsigned long long *slots;     // see kGmMidSlotBytes
    const void *dinv;              // PRE: the Jacobi preconditioner's diagonal, w = dinv .* (A v) (left preconditioning, TSL:351)
    hipk_gm_scal *scal;
    double eps;
    int test_not_resident, slot_stride, xcd_aware;
};
static inline size_t hipk_gm_mid_lds_bytes(int win, size_t sv = 8) {   // sv: bytes of a vector element
    return (size_t)(win + 8 + HIPK_BASE_CHUNK) * sv + (size_t)(kGmMidCols * 256 + 6 * 40 + 32 + 8 + kMidPlanSlots / 2) * sizeof(double);
}

// Diagnostic twin (make stamps): thread 0 of every workgroup sums, over the steps of a launch, the constant 100 MHz clock between
// its phase boundaries; tools/gmres_mid_stamps_probe.py prints where a cycle goes.
#ifdef HIPK_GM_STAMPS
#define HIPK_GMM_NSTAMP 14
__device__ unsigned long long hipk_gm_mid_stamps[kGmMidMaxChunks * HIPK_GMM_NSTAMP];
#undef HIPK_MSTAMP
#define HIPK_MSTAMP(k)                                                  \
    do {                                                                \
        const unsigned long long t_ = __builtin_amdgcn_s_memrealtime(); \
        st_acc[k] += t_ - st_prev;                                      \
        st_prev = t_;                                                   \
    } while (0)
#else
#undef HIPK_MSTAMP
#define HIPK_MSTAMP(k)
#endif

// PRE: Jacobi-preconditioned GMRES -- the row scaling of the SpMV kernels' epilogue (HIPK_SPMV_SCALE: out = dinv .* out before the
// fused ||w||^2), which is the whole difference inside a cycle (hipk_pgmres_solve).
// T: the handle's dtype (basis, window, w / q and the element-wise arithmetic in T; dots, h and the small arrays in double).
template <typename T, int W, bool PRE = false>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4, 4))) void hipk_gm_mid_kernel(hipk_gm_mid_args a) {
    constexpr int NTHR = 1024, CH = HIPK_BASE_CHUNK, R = CH / NTHR, TSTEP = NTHR / HIPK_TILE;
    extern __shared__ double mid_lds[];
    const int g = a.g, WIN_ = a.win, m = a.m;
    const int wg = a.xcd_aware ? hipk_xcd_chunk(blockIdx.x, g) : ((int)blockIdx.x < g ? (int)blockIdx.x : -1);
    if (wg < 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, tw = wave & 3, tl = tid & (HIPK_TILE - 1), t0 = tid >> 8;
    const int q4 = tid >> 8, t = tid & 255;   // column group / virtual thread of the vector phases
    constexpr int VEC = hipk_vec<T>::VEC;
    T *vw = (T *)mid_lds;                    // v_k at the window's columns; vw[WIN] = 0.0 for the padding entries of short rows
    T *wq = vw + WIN_ + 8;                   // w, then q, of the own rows
    double *sbm = (double *)(wq + CH);       // [32][256] per-column fold buffers
    double *hs = sbm + kGmMidCols * 256;     // [40] h of the pass
    double *rvec = hs + 40;                  // [40] h accumulated over the passes (TSL:302-305)
    double *hc = rvec + 40;                  // [40] column k of H under the Givens rotations
    double *gvs = hc + 40;                   // [80] the rotations (cs, sn)
    double *bvs = gvs + 80;                  // [40] beta_vec under the rotations (TSL:595-623)
    double *ts = bvs + 40;                   // [32] wavefront sums of w .* w
    int *flags = (int *)(ts + 32);           // [0] a hand-off failed, [1] stop after this step
    int *stile = flags + 16;                 // the window's tiles: slot s holds columns 256 stile[s] .. + 255
    const int64_t n = a.n, base = (int64_t)wg * CH;
    const int tlo = a.plan.tlo[wg], WINc = a.plan.nslot[wg] * HIPK_TILE;   // this workgroup's window
    const short *tmap = a.plan.map + (size_t)wg * kMidPlanRange;
    const int H_ = __builtin_amdgcn_readfirstlane((int)tmap[(int)(base >> 8) - tlo] * HIPK_TILE);   // where the own tiles sit in the window
    if (tid < kMidPlanSlots) stile[tid] = (tid * HIPK_TILE < WINc) ? a.plan.tiles[wg * kMidPlanSlots + tid] : 0;
    __syncthreads();
    const int ntiles = (int)((n + HIPK_TILE - 1) / HIPK_TILE);
    hipk_gm_scal *scal = a.scal;
    const hipk_gm_view gv_ = scal->v;        // the header block's arrays (workgroup 0 mirrors its private copies there)
    const int ldh = gv_.ldh;
    const int ss = a.slot_stride;
    // ONE buffer resource over v_ll .. the end of the slot arrays (both lie in the solve's work buffer, v_ll first)
    const unsigned s_off = (unsigned)((const char *)a.slots - (const char *)a.v_ll);
    const hipk_ll_rsrc ll = hipk_ll_make(a.v_ll, (size_t)s_off + kGmMidSlotBytes);
    const unsigned col_bytes = (unsigned)g * (unsigned)ss * 16u;            // one column's slot array
    const unsigned qq_o = s_off + kGmMidCols * col_bytes, ww_o0 = qq_o + col_bytes;
    if (tid < 2) flags[tid] = 0;

    // ---- the own rows' matrix entries in registers (thread: rows 256 (t0 + 4 k) + tl); v_0 over the window from the basis
    T vj[R][W], dj[R];
    T *const Vb = (T *)a.V;
    int cj[R][W];
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int64_t row = base + (t0 + TSTEP * k) * HIPK_TILE + tl;
        dj[k] = (PRE && row < n) ? ((const T *)a.dinv)[row] : (T)1;
        int lo = 0, len = 0;
        if (row < n) {
            lo = a.crow[row];
            len = a.crow[row + 1] - lo;
        }
#pragma unroll
        for (int j = 0; j < W; ++j) {   // short rows padded with (+0.0, the zero slot WIN): see hipk_cg_mid.h
            const bool has = j < len;
            const int cc = has ? a.col[lo + j] : 0;
            cj[k][j] = has ? (int)tmap[(cc >> 8) - tlo] * HIPK_TILE + (cc & (HIPK_TILE - 1)) : WIN_;
            vj[k][j] = has ? ((const T *)a.val)[lo + j] : (T)0;
        }
    }
    if (tid < 8) vw[WIN_ + tid] = (T)0;
    for (int idx = tid; idx < WINc; idx += NTHR) {
        const int64_t gc = (int64_t)stile[idx >> 8] * HIPK_TILE + (idx & (HIPK_TILE - 1));
        vw[idx] = gc < n ? Vb[gc] : (T)0;
    }
    if (tid < 40) {
        hs[tid] = 0.0;
        rvec[tid] = 0.0;
        hc[tid] = 0.0;
        bvs[tid] = (tid <= ldh && tid < 34) ? gv_.beta_vec[tid] : 0.0;
    }
    if (tid < 80) gvs[tid] = (tid < 2 * ldh && tid < 64) ? gv_.gv[tid] : 0.0;
    const int incremental = scal->incremental;
    const double ptol = scal->ptol, eps = a.eps;
    // rows whose v another workgroup's window holds (their tile is in that window's list): element i = 2t + 512 q4 (+1) of the chunk
    const bool pub = a.plan.needed[(int)(base >> 8) + ((2 * t + 512 * q4) >> 8)] != 0;

    // every workgroup resident?  Nothing has been modified yet: a failure leaves the cycle to the launches
    int epoch = 0;
    if (!hipk_gbar(&scal->bar, g, epoch, flags) || a.test_not_resident) {
        if (tid == 0) scal->redo = -1;
        return;
    }
#define HIPK_GMM_FAIL()                  \
    if (flags[0]) {                      \
        if (tid == 0) scal->redo = -3;   \
        return;                          \
    }
#ifdef HIPK_GM_STAMPS
    unsigned long long st_acc[HIPK_GMM_NSTAMP] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_prev = __builtin_amdgcn_s_memrealtime();
#endif
    for (int k = 0; k < m; ++k) {
        // the window geometry is made opaque once per step (hipk_bi_mid.h: LDS addresses formed ahead of the loop cost registers)
        int H = H_, WIN = WIN_;
        asm volatile("" : "+s"(H), "+s"(WIN));
        T *wq = vw + WIN + 8;
        double *sbm = (double *)(wq + CH);
        const unsigned seq_k = (unsigned)k + 1u;
        const unsigned ww_o = ww_o0 + (unsigned)(k & 1) * col_bytes;
        // ---- w = A v_k (own rows; products rounded, added in CSR order), wavefront sums of w .* w   (TSL:351-352)
        {
            double d[R];
#pragma unroll
            for (int kk = 0; kk < R; ++kk) {
                const int lrow = (t0 + TSTEP * kk) * HIPK_TILE + tl;
                T acc = (T)0;
#pragma unroll
                for (int j = 0; j < W; ++j) {
                    const T pr = vj[kk][j] * vw[cj[kk][j]];
                    acc = acc + pr;
                }
                if (PRE) acc = dj[kk] * acc;   // w = M (A v_k)
                wq[lrow] = acc;   // rows beyond n: padding only, +0.0
                d[kk] = (base + lrow < n) ? (double)acc * (double)acc : 0.0;
            }
            const double s2 = hipk_wave_sum_pair(d[0], d[1]);
            if ((lane & 31) == 0) ts[(t0 + TSTEP * (lane >> 5)) * 4 + tw] = s2;
        }
        __syncthreads();
        HIPK_MSTAMP(0);
        if (tid < 64) {
            const double part = hipk_mid_tiles_fold(ts, lane, wg * (CH / HIPK_TILE), ntiles);
            if (lane == 0) hipk_ll_put(ll, wg * ss, part, seq_k, ww_o);
        }
        double qq = 0.0, ww = 0.0;
        for (int pass = 0; pass < 2; ++pass) {
            if (pass == 1) {   // second CGS pass iff ||r|| < ||q|| / sqrt(2)  (hipk_gm_decide_kernel, hipk_gm_want_pass2): every
                               // wavefront for itself -- lane j holds rvec[j], the chain of the spec takes them by v_readlane (as a loop
                               // of dependent LDS reads it cost 1 us per step)
                const double rl = (lane <= k) ? rvec[lane] : 0.0;
                double rr = 0.0;
                for (int j = 0; j <= k; ++j) {
                    const double rj = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(rl), j), __builtin_amdgcn_readlane(__double2loint(rl), j));
                    rr = fma(rj, rj, rr);
                }
                double qnorm = sqrt(qq < 0.0 ? 0.0 : qq);
                if (!(qnorm > eps)) qnorm = 0.0;
                double rnorm = sqrt(rr < 0.0 ? 0.0 : rr);
                if (!(rnorm > eps)) rnorm = 0.0;
                if (!(rnorm < qnorm * HIPK_INV_SQRT2)) break;
            }
            const unsigned seq_p = 2u * (unsigned)k + (unsigned)pass + 1u;
            // ---- h_j = <V_j, w>, j <= k: chains of the own chunk (column group q4 takes j = q4, q4 + 4, ...; virtual thread t
            // the elements {2t, 2t+1} + 512 jj ascending), chunk trees, partials out   (hipk_gm_multidot_stream_kernel)
            {
                // virtual thread t: elements {VEC t .. VEC t + VEC - 1} + 256 VEC jj ascending
                constexpr int NJ = CH / (256 * VEC);
                double wv[NJ][VEC];
#pragma unroll
                for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
                    for (int e = 0; e < VEC; ++e) wv[jj][e] = (double)wq[VEC * t + 256 * VEC * jj + e];
                for (int j = q4; j <= k; j += 4) {
                    const T *Vj = Vb + (int64_t)j * a.ldv;
                    T vv[NJ][VEC];
#pragma unroll
                    for (int jj = 0; jj < NJ; ++jj) {
                        const int64_t i = base + VEC * t + 256 * VEC * jj;
                        const int64_t left = n - i;
                        hipk_ld<T>(Vj, i, left >= VEC ? VEC : (left > 0 ? (int)left : 0), vv[jj]);
                    }
                    double acc = 0.0;
#pragma unroll
                    for (int jj = 0; jj < NJ; ++jj) {
                        const int64_t i = base + VEC * t + 256 * VEC * jj;
#pragma unroll
                        for (int e = 0; e < VEC; ++e)
                            if (i + e < n) acc = fma((double)vv[jj][e], wv[jj][e], acc);
                    }
                    sbm[j * 256 + t] = acc;
                }
            }
            __syncthreads();
        HIPK_MSTAMP(1);
            for (int j = wave; j <= k; j += NTHR / 64) {
                const double part = hipk_mid_tree(sbm + j * 256, lane);
                if (lane == 0) hipk_ll_put(ll, wg * ss, part, seq_p, s_off + (unsigned)j * col_bytes);
            }
            __syncthreads();
        HIPK_MSTAMP(2);
            // hand-off: every workgroup folds the g partials of every column in the spec's order   (hipk_gm_hreduce_kernel)
            {
                constexpr int NB = 4;   // columns of a thread in flight (8 spill registers)
                for (int j0 = q4; j0 <= k; j0 += 4 * NB) {
                    hipk_v4u pw_[NB];
#pragma unroll
                    for (int b = 0; b < NB; ++b)
                        if (j0 + 4 * b <= k && t < g) pw_[b] = hipk_ll_load(ll, t * ss, s_off + (unsigned)(j0 + 4 * b) * col_bytes);
#pragma unroll
                    for (int b = 0; b < NB; ++b)
                        if (j0 + 4 * b <= k) {
                            double acc = 0.0;
                            if (t < g) {
                                double v = 0.0;
                                if (!hipk_ll_wait(ll, t * ss, seq_p, pw_[b], v, s_off + (unsigned)(j0 + 4 * b) * col_bytes)) flags[0] = 1;
                                acc = acc + v;
                            }
                            sbm[(j0 + 4 * b) * 256 + t] = acc;
                        }
                }
            }
            __syncthreads();
        HIPK_MSTAMP(3);
            HIPK_GMM_FAIL()
            for (int j = wave; j <= k; j += NTHR / 64) {
                const double h = hipk_mid_tree(sbm + j * 256, lane);
                if (lane == 0) hs[j] = h;
            }
            __syncthreads();
        HIPK_MSTAMP(4);
            // ---- q = w - V h (element i = 2t + 512 q4 + e of the own chunk), rvec += h   (hipk_gm_update_stream_kernel)
            {
                const int i = 2 * t + 512 * q4;
                double s0 = 0.0, s1 = 0.0;
                for (int j = 0; j <= k; ++j) {
                    const T *Vj = Vb + (int64_t)j * a.ldv + base;
                    T v0 = (T)0, v1 = (T)0;
                    if (base + i + 1 < n) {
                        if constexpr (sizeof(T) == 8) {
                            const double2 vv = *(const double2 *)(Vj + i);
                            v0 = vv.x;
                            v1 = vv.y;
                        } else {
                            const float2 vv = *(const float2 *)(Vj + i);
                            v0 = vv.x;
                            v1 = vv.y;
                        }
                    } else if (base + i < n) {
                        v0 = Vj[i];
                    }
                    const double hj = hs[j];
                    s0 = fma((double)v0, hj, s0);
                    s1 = fma((double)v1, hj, s1);
                }
                wq[i] = (T)((double)wq[i] - s0);
                wq[i + 1] = (T)((double)wq[i + 1] - s1);
                if (tid <= k) rvec[tid] = ((pass == 0) ? 0.0 : rvec[tid]) + hs[tid];
            }
            __syncthreads();
        HIPK_MSTAMP(5);
            if (tid < 256) {
                double acc = 0.0;
#pragma unroll
                for (int jj = 0; jj < CH / (256 * VEC); ++jj)
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        const int i = VEC * tid + 256 * VEC * jj + e;
                        const double v = (double)wq[i];
                        if (base + i < n) acc = fma(v, v, acc);
                    }
                sbm[tid] = acc;
            }
            __syncthreads();
        HIPK_MSTAMP(6);
            if (tid < 64) {
                const double part = hipk_mid_tree(sbm, lane);
                if (lane == 0) hipk_ll_put(ll, wg * ss, part, seq_p, qq_o);
            }
            // hand-off: the partials of <q,q> (and, once per step, of ||A v||^2)
            if (tid < 256) {
                double a0 = 0.0, a1 = 0.0;
                if (tid < g) {
                    const hipk_v4u w0_ = hipk_ll_load(ll, tid * ss, qq_o);
                    hipk_v4u w1_ = w0_;
                    if (pass == 0) w1_ = hipk_ll_load(ll, tid * ss, ww_o);
                    double v0 = 0.0, v1 = 0.0;
                    if (!hipk_ll_wait(ll, tid * ss, seq_p, w0_, v0, qq_o)) flags[0] = 1;
                    if (pass == 0 && !hipk_ll_wait(ll, tid * ss, seq_k, w1_, v1, ww_o)) flags[0] = 1;
                    a0 = a0 + v0;
                    a1 = a1 + v1;
                }
                sbm[256 + tid] = a0;
                if (pass == 0) sbm[512 + tid] = a1;
            }
            __syncthreads();
            HIPK_GMM_FAIL()
        HIPK_MSTAMP(7);
            qq = hipk_mid_tree(sbm + 256, lane);
            if (pass == 0) ww = hipk_mid_tree(sbm + 512, lane);
            __syncthreads();
        HIPK_MSTAMP(8);
        }
        // ---- v_{k+1} = q / ||q|| (zero when ||q|| <= eps ||A v_k||), column k of H, breakdown, Givens + early exit
        // (hipk_gm_normalize_kernel)
        HIPK_MSTAMP(11);   // (the second-pass decision of a step that needs none)
        double norm1 = sqrt(qq < 0.0 ? 0.0 : qq);
        double norm0 = sqrt(ww < 0.0 ? 0.0 : ww);
        if (!(norm0 > eps)) norm0 = 0.0;
        const double thr = eps * norm0;
        const bool use = norm1 > thr;
        const T nrm = (T)norm1;
        const unsigned seq_v = (unsigned)k + 1u;
        {
            const int i = 2 * t + 512 * q4;
            T *Vn = Vb + (int64_t)(k + 1) * a.ldv + base;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const T v = use ? wq[i + e] / nrm : (T)0;
                vw[H + i + e] = v;
                if (base + i + e < n) {
                    Vn[i + e] = v;
                    if (pub) hipk_ll_put(ll, (unsigned)(base + i + e), (double)v, seq_v);   // (a float travels as the double it equals)
                }
            }
        }
        HIPK_MSTAMP(12);   // (scaling, stores)
        if (wg == 0 && tid <= k) gv_.H[tid * ldh + k] = rvec[tid];   // column k of H, the header block's copy
        if (incremental) {   // column k of H for the rotations: one element per thread (as one thread's loop of dependent LDS
                             // round trips the copy cost ~1 us per step)
            if (tid <= k) hc[tid] = rvec[tid];
            __syncthreads();
        }
        if (tid == 0) {
            if (!use) norm1 = 0.0;
            hc[k + 1] = norm1;
            bool stop = false;
            int breakdown = 0;
            if (norm1 == 0.0) {  // TSL:387
                breakdown = 1;
                stop = true;
            }
            if (wg == 0) gv_.H[(k + 1) * ldh + k] = norm1;
            double err = 0.0;
            if (incremental) {
                for (int i = 0; i < k; ++i) {
                    const double cs = gvs[2 * i], sn = gvs[2 * i + 1];
                    const double p0 = cs * hc[i], p1 = sn * hc[i + 1];
                    const double t0_ = p0 - p1;
                    const double p2 = sn * hc[i], p3 = cs * hc[i + 1];
                    hc[i + 1] = p2 + p3;
                    hc[i] = t0_;
                }
                double cs, sn;
                hipk_givens(hc[k], hc[k + 1], cs, sn);
                gvs[2 * k] = cs;
                gvs[2 * k + 1] = sn;
                {
                    const double p0 = cs * hc[k], p1 = sn * hc[k + 1];
                    hc[k] = p0 - p1;
                }
                hc[k + 1] = 0.0;
                const double p0 = cs * bvs[k], p1 = sn * bvs[k + 1];
                const double t0_ = p0 - p1;
                const double p2 = sn * bvs[k], p3 = cs * bvs[k + 1];
                bvs[k + 1] = p2 + p3;
                bvs[k] = t0_;
                err = fabs(bvs[k + 1]);
                if (!(err > ptol)) stop = true;  // TSL:591
                if (wg == 0) {
                    gv_.gv[2 * k] = cs;
                    gv_.gv[2 * k + 1] = sn;
                    for (int j = 0; j <= k; ++j) gv_.R[j * ldh + k] = hc[j];
                    gv_.beta_vec[k] = bvs[k];
                    gv_.beta_vec[k + 1] = bvs[k + 1];
                    scal->err = err;
                }
            }
            if (wg == 0) {
                scal->steps_done = k + 1;
                if (breakdown) scal->breakdown = 1;
                if (stop) scal->stop_step = k + 1;
            }
            flags[1] = stop ? 1 : 0;
        }
        __syncthreads();
        HIPK_MSTAMP(9);
        if (flags[1] || k + 1 >= m) break;
        // hand-off: v_{k+1} at the window's halo columns
        for (int idx = tid; idx < WINc - CH; idx += NTHR) {   // the window without the own tiles (contiguous at H)
            const int widx = idx < H ? idx : idx + CH;
            const int64_t gc = (int64_t)stile[widx >> 8] * HIPK_TILE + (widx & (HIPK_TILE - 1));
            if (gc < n) {
                double v = 0.0;
                if (!hipk_ll_wait(ll, (unsigned)gc, seq_v, hipk_ll_load(ll, (unsigned)gc), v)) flags[0] = 1;
                vw[widx] = (T)v;
            }
        }
        __syncthreads();
        HIPK_GMM_FAIL()
        HIPK_MSTAMP(10);
    }
#undef HIPK_GMM_FAIL
#ifdef HIPK_GM_STAMPS
    if (tid == 0)
        for (int j = 0; j < HIPK_GMM_NSTAMP; ++j) hipk_gm_mid_stamps[wg * HIPK_GMM_NSTAMP + j] = st_acc[j];
#endif
}
#endif  // HIPK_GM_MID_H
