// hipk_bi_mid.h -- BiCGStab for launch-bound MID-SIZE systems (9 .. 256 reduction chunks, fp64, M = identity, banded rows of
// <= 12 entries): THE WHOLE LOOP IN ONE LAUNCH, one 1024-thread workgroup per reduction chunk -- the scheme of hipk_cg_mid.h
// (flagged 16-byte words, LDS windows, every workgroup folds all chunk partials itself) with three hand-offs per iteration
// instead of five launches:
//   K1 tests, beta, p = r + beta (p - omega q) over the WINDOW | K2 q = A p (own rows), tile sums of rhat .* q
//        -> hand-off 1: the chunk partial of <rhat,q> + q (for the neighbours' windows)
//   K3 alpha test, s = r - alpha q over the window (in place of r), <s,s> | K4 t = A s (own rows), tile sums of s .* t, t .* t
//        -> hand-off 2: the chunk partials of <s,s>, <t,s>, <t,t>
//   K5 omega tests, x, r (own rows), <r,r>, <rhat,r>
//        -> hand-off 3: those two partials + r
// p and s are never exchanged: each workgroup advances them over its whole window from the exchanged r and q (the owners'
// formulas on the owners' operands, the same bits).  Arithmetic and every scalar test (TSL:894-936, 961) as in the five
// kernels of hipk_bicgstab.hip, bit for bit; tests/test_gpu_api.py::test_bicgstab_mid_one_launch_is_bit_identical.
#ifndef HIPK_BI_MID_H
#define HIPK_BI_MID_H
#include "hipk_mid.h"

static constexpr int kBiMidMaxChunks = 256;   // one chunk per workgroup, one workgroup per CU
static constexpr int kBiMidKinds = 6;   // chunk-partial slot arrays: <rhat,q> | <s,s> <t,s> <t,t> | <r,r> <rhat,r>
static constexpr size_t kBiMidSlotBytes = (size_t)kBiMidKinds * kMidMaxChunks * 256;

struct hipk_bi_mid_args {
    int64_t n;
    int g, win;                        // chunks; doubles of a window in LDS = 256 x the most tiles any workgroup's window holds
    hipk_mid_plan plan;                // which tiles (hipk_mid.h)
    const int *crow, *col;
    const void *val;                   // values, x, r, p, q, rhat, dinv: of the handle's dtype (the kernel's T)
    void *x, *r, *p, *q;
    const void *rhat;
    const void *dinv;                // PRE: Jacobi preconditioning, M = diag(dinv) applied BEFORE A (TSL:908, 922)
    unsigned long long *q_ll, *r_ll;   // [2 n] flagged words of q and r
    unsigned long long *slots;         // kBiMidKinds arrays of g slots, slot_stride 16-byte words apart
    double *part_rr, *part_rhr;        // chunk partials in memory: read by the first iteration of a launch, left by its last
    hipk_bi_scal *scal;
    int64_t it0, maxiter, max_its;
    int test_not_resident;
    int slot_stride;
    int xcd_aware;
};
static inline size_t hipk_bi_mid_lds_bytes(int win, bool pre = false, size_t sv = 8) {   // pre: + the window of dinv; sv: bytes of a vector element
    return (size_t)((pre ? 4 : 3) * win + 24 + HIPK_BASE_CHUNK) * sv + (size_t)(2 * 3 * 256 + 3 * 32 + 8 + kMidPlanSlots / 2) * sizeof(double);
}

// PRE: phat = dinv .* p and shat = dinv .* s are the products' inputs (formed at the gathered columns from a fourth LDS window, the
// owners' formulas on the owners' operands) and x advances with them (TSL:908, 922, 942): hipk_bi_*_kernel<T, true> bit for bit.
// T: the handle's dtype (vectors, windows, element-wise arithmetic in T; dots and scalars in double).
template <typename T, int W, bool PRE = false>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4, 4))) void hipk_bi_mid_kernel(hipk_bi_mid_args a) {
    constexpr int NTHR = 1024, CH = HIPK_BASE_CHUNK, R = CH / NTHR, TSTEP = NTHR / HIPK_TILE;
    constexpr double EPS = hipk_eps<T>::v;
    constexpr int VEC = hipk_vec<T>::VEC;
    extern __shared__ double mid_lds[];
    const int g = a.g, WIN = a.win;
    const int wg = a.xcd_aware ? hipk_xcd_chunk(blockIdx.x, g) : ((int)blockIdx.x < g ? (int)blockIdx.x : -1);
    if (wg < 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, tw = wave & 3, tl = tid & (HIPK_TILE - 1), t0 = tid >> 8;
    T *pw = (T *)mid_lds;            // p at the window's columns; pw[WIN] = 0.0 for the padding entries of short rows
    T *qw = pw + WIN + 8;            // q at the window's columns
    T *rw = qw + WIN;                // r, between K3 and K5 s, at the window's columns; rw[WIN] = 0.0 as in pw
    T *hw = rw + WIN + 8;            // rhat of the own rows (the <rhat,r> chains read it in the spec's virtual-thread layout)
    double *sb = (double *)(hw + CH);   // 2 x [3 x 256]: fold buffers, used alternately
    double *ts = sb + 2 * 3 * 256;   // [3 x 32] wavefront sums of the tiled dots
    int *fail = (int *)(ts + 3 * 32);
    T *dw = (T *)(ts + 3 * 32 + 1);  // PRE: dinv at the window's columns; dw[WIN] = 0.0
    int *stile = (int *)(dw + (PRE ? WIN + 8 : 0));   // the window's tiles: slot s holds columns 256 stile[s] .. + 255
    const int64_t n = a.n, base = (int64_t)wg * CH;
    const int tlo = a.plan.tlo[wg], WINc = a.plan.nslot[wg] * HIPK_TILE;   // this workgroup's window
    const short *tmap = a.plan.map + (size_t)wg * kMidPlanRange;
    const int H = __builtin_amdgcn_readfirstlane((int)tmap[(int)(base >> 8) - tlo] * HIPK_TILE);   // where the own tiles sit in the window
    if (tid < kMidPlanSlots) stile[tid] = (tid * HIPK_TILE < WINc) ? a.plan.tiles[wg * kMidPlanSlots + tid] : 0;
    __syncthreads();
    const int ntiles = (int)((n + HIPK_TILE - 1) / HIPK_TILE);
    hipk_bi_scal *scal = a.scal;
    if (tid == 0) *fail = 0;
    const int ss = a.slot_stride;
    // ONE buffer resource over q_ll .. the end of the slot arrays (they all lie in the solve's work buffer, q_ll first); the arrays
    // are told apart by a scalar byte offset (eight resources cost 32 scalar registers the loop does not have)
    const unsigned r_off = (unsigned)((const char *)a.r_ll - (const char *)a.q_ll), s_off = (unsigned)((const char *)a.slots - (const char *)a.q_ll);
    constexpr unsigned KB = (unsigned)kMidMaxChunks * 256u;   // bytes of one kind's slot array
    const hipk_ll_rsrc ll = hipk_ll_make(a.q_ll, (size_t)s_off + kBiMidSlotBytes);
    const unsigned rq_o = s_off, ss_o = s_off + KB, ts_o = s_off + 2 * KB, tt_o = s_off + 3 * KB, rr_o = s_off + 4 * KB, rhr_o = s_off + 5 * KB;

    // ---- the own rows: x and the matrix entries in registers (thread t: rows 256 (t0 + 4 k) + tl); r (s), q, p of the own rows are
    // the [H, H + CH) parts of the LDS windows, rhat sits in hw -- the register file is spent on the matrix (128 VGPRs at 4 waves per SIMD)
    T xo[R], vj[R][W];
    int cj[R][W];
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int lrow = (t0 + TSTEP * k) * HIPK_TILE + tl;
        const int64_t row = base + lrow;
        const bool live = row < n;
        xo[k] = live ? ((const T *)a.x)[row] : (T)0;
        hw[lrow] = live ? ((const T *)a.rhat)[row] : (T)0;
        int lo = 0, len = 0;
        if (live) {
            lo = a.crow[row];
            len = a.crow[row + 1] - lo;
        }
#pragma unroll
        for (int j = 0; j < W; ++j) {   // short rows padded with (+0.0, the zero slot WIN): see hipk_cg_mid.h
            const bool has = j < len;
            const int cc = has ? a.col[lo + j] : 0;
            cj[k][j] = has ? (int)tmap[(cc >> 8) - tlo] * HIPK_TILE + (cc & (HIPK_TILE - 1)) : WIN;
            vj[k][j] = has ? ((const T *)a.val)[lo + j] : (T)0;
        }
    }
    if (tid < 8) pw[WIN + tid] = (T)0;
    for (int idx = tid; idx < WINc; idx += NTHR) {
        const int64_t gc = (int64_t)stile[idx >> 8] * HIPK_TILE + (idx & (HIPK_TILE - 1));
        const bool in = gc < n;
        pw[idx] = in ? ((const T *)a.p)[gc] : (T)0;
        qw[idx] = in ? ((const T *)a.q)[gc] : (T)0;
        rw[idx] = in ? ((const T *)a.r)[gc] : (T)0;
        if (PRE) dw[idx] = in ? ((const T *)a.dinv)[gc] : (T)0;
    }
    if (PRE && tid < 8) dw[WIN + tid] = (T)0;
    if (tid < 8) rw[WIN + tid] = (T)0;   // s at the zero slot (the second product gathers from rw)
    double rho = scal->rho, alpha = scal->alpha, omega = scal->omega;
    const double atol2 = scal->atol2;
    const int64_t stop0 = scal->stop_it;
    double rs_last = scal->rs_last;
    bool pub[R];   // rows whose q / r another workgroup's window holds (their tile is in that window's list)
#pragma unroll
    for (int k = 0; k < R; ++k) pub[k] = a.plan.needed[(int)(base >> 8) + t0 + TSTEP * k] != 0;

    // every workgroup resident?  Nothing has been modified yet: a failure leaves the solve to the launch sequence
    int epoch = 0;
    if (!hipk_gbar(&scal->bar, g, epoch, fail) || a.test_not_resident) {
        if (tid == 0) scal->redo = -1;
        return;
    }
#define HIPK_BIM_FAIL()                       \
    if (*fail) {                              \
        if (tid == 0) scal->redo = -3;        \
        return;                               \
    }
    unsigned seq = 0;
    int buf = 0;
    int64_t it = a.it0, iters = a.it0, stop_it = stop0;
    int code = 0, extra_mv = 0;
    double part_rr_own = 0.0, part_rhr_own = 0.0;   // this chunk's partials of the last completed iteration
    bool have_parts = false;
    const int H_ = H, WIN_ = WIN;
    while (it < stop_it && it - a.it0 < a.max_its) {
        ++seq;
        // the window geometry is made opaque once per iteration: otherwise every LDS address of the loop (three windows x gathered
        // columns, own rows, passes) is formed ahead of it and kept in a register -- 25 VGPRs spilled at W = 5 with that
        int H = H_, WIN = WIN_;
        asm volatile("" : "+s"(H), "+s"(WIN));
        T *qw = pw + WIN + 8, *rw = qw + WIN, *hw = rw + WIN + 8;
        T *dw = (T *)((double *)(hw + CH) + 2 * 3 * 256 + 3 * 32 + 1);
        double *sbb = sb + buf * 3 * 256;
        // ---- K1: rs = <r,r>, rho' = <rhat,r> -> tests; beta; p = r + beta (p - omega q) over the window   (TSL:893-907)
        if (tid < 256) {
            double a0 = 0.0, a1 = 0.0;
            if (seq == 1) {   // first iteration of the launch: the chunk partials the launches before this one left in memory
                if (tid < g) {   // g <= 256: one partial per thread
                    a0 = a0 + a.part_rr[tid];
                    a1 = a1 + a.part_rhr[tid];
                }
            } else {
                hipk_mid_poll2<1>(ll, rr_o, rhr_o, g, seq - 1, fail, ss, a0, a1);
            }
            sbb[tid] = a0;
            sbb[256 + tid] = a1;
        }
        __syncthreads();
        const double rs = hipk_mid_tree(sbb, lane), rho_new = hipk_mid_tree(sbb + 256, lane);
        buf ^= 1;
        HIPK_BIM_FAIL()
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
        for (int idx = tid; idx < WINc; idx += NTHR) {       // TSL:907
            const T t1 = om * qw[idx];
            const T t2 = pw[idx] - t1;
            const T t3 = beta * t2;
            pw[idx] = rw[idx] + t3;
        }
        __syncthreads();
        // ---- K2: q = A p (own rows), wavefront sums of rhat .* q   (TSL:909-910)
        {
            double d[R];
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const int lrow = (t0 + TSTEP * k) * HIPK_TILE + tl;
                T acc = (T)0;
#pragma unroll
                for (int j = 0; j < W; ++j) {
                    const T pin = PRE ? dw[cj[k][j]] * pw[cj[k][j]] : pw[cj[k][j]];   // phat = M p (TSL:908)
                    const T pr = vj[k][j] * pin;
                    acc = acc + pr;
                }
                qw[H + lrow] = acc;
                d[k] = (base + lrow < n) ? (double)hw[lrow] * (double)acc : 0.0;
                if (base + lrow < n && pub[k]) hipk_ll_put(ll, (unsigned)(base + lrow), (double)acc, seq);   // (a float travels as the double it equals)
            }
            const double s2 = hipk_wave_sum_pair(d[0], d[1]);
            if ((lane & 31) == 0) ts[(t0 + TSTEP * (lane >> 5)) * 4 + tw] = s2;
        }
        __syncthreads();
        if (tid < 64) {
            const double part = hipk_mid_tiles_fold(ts, lane, wg * (CH / HIPK_TILE), ntiles);
            if (lane == 0) hipk_ll_put(ll, wg * ss, part, seq, rq_o);
        }
        // hand-off 1: q at the window's halo columns, the partials of <rhat,q>
        for (int idx = tid; idx < WINc - CH; idx += NTHR) {   // the window without the own tiles (contiguous at H)
            const int widx = idx < H ? idx : idx + CH;
            const int64_t gc = (int64_t)stile[widx >> 8] * HIPK_TILE + (widx & (HIPK_TILE - 1));
            if (gc < n) {
                double v = 0.0;
                if (!hipk_ll_wait(ll, (unsigned)gc, seq, hipk_ll_load(ll, (unsigned)gc), v)) *fail = 1;
                qw[widx] = (T)v;
            }
        }
        sbb = sb + buf * 3 * 256;
        if (tid < 256) sbb[tid] = hipk_mid_poll<1>(ll, g, seq, fail, ss, rq_o);
        __syncthreads();
        const double rq = hipk_mid_tree(sbb, lane);
        buf ^= 1;
        HIPK_BIM_FAIL()
        // ---- K3: alpha' = rho'/<rhat,q> -> test; s = r - alpha' q over the window (in place of r); <s,s>   (TSL:910-920)
        const double alpha_new = rho_new / rq;  // TSL:910
        if (fabs(alpha_new) < EPS) {            // TSL:913-915
            stop_it = it;
            code = -11;
            extra_mv = 1;
            break;
        }
        const T al = (T)alpha_new;
        for (int idx = tid; idx < WINc; idx += NTHR) {   // TSL:917
            const T m = al * qw[idx];
            rw[idx] = rw[idx] - m;
        }
        __syncthreads();
        sbb = sb + buf * 3 * 256;
        if (tid < 256) {
            double acc = 0.0;   // virtual thread t of the chunk: elements {VEC t .. VEC t + VEC - 1} + 256 VEC j ascending
#pragma unroll
            for (int j = 0; j < CH / (256 * VEC); ++j)
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    const int i = VEC * tid + 256 * VEC * j + k;
                    const double v = (double)rw[H + i];
                    if (base + i < n) acc = fma(v, v, acc);
                }
            sbb[tid] = acc;
        }
        // ---- K4: t = A s (own rows), wavefront sums of s .* t and t .* t   (TSL:923-930)
        T to[R];
        {
            double d0[R], d1[R];
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const int lrow = (t0 + TSTEP * k) * HIPK_TILE + tl;
                T acc = (T)0;
#pragma unroll
                for (int j = 0; j < W; ++j) {
                    const T sin_ = PRE ? dw[cj[k][j]] * rw[cj[k][j]] : rw[cj[k][j]];   // shat = M s (TSL:922)
                    const T pr = vj[k][j] * sin_;
                    acc = acc + pr;
                }
                to[k] = acc;
                const bool live = base + lrow < n;
                d0[k] = live ? (double)rw[H + lrow] * (double)acc : 0.0;   // s of the own row (K3's pass over the window)
                d1[k] = live ? (double)acc * (double)acc : 0.0;
            }
            const double s0 = hipk_wave_sum_pair(d0[0], d0[1]), s1 = hipk_wave_sum_pair(d1[0], d1[1]);
            if ((lane & 31) == 0) {
                ts[32 + (t0 + TSTEP * (lane >> 5)) * 4 + tw] = s0;
                ts[64 + (t0 + TSTEP * (lane >> 5)) * 4 + tw] = s1;
            }
        }
        __syncthreads();
        if (tid < 192) {   // wavefront 0: <s,s> of the chunk; 1: <t,s>; 2: <t,t>
            double part;
            if (wave == 0) part = hipk_mid_tree(sbb, lane);
            else part = hipk_mid_tiles_fold(ts + 32 * wave, lane, wg * (CH / HIPK_TILE), ntiles);
            if (lane == 0) hipk_ll_put(ll, wg * ss, part, seq, wave == 0 ? ss_o : wave == 1 ? ts_o : tt_o);
        }
        buf ^= 1;
        // hand-off 2: the partials of <s,s>, <t,s>, <t,t>
        sbb = sb + buf * 3 * 256;
        if (tid < 256) {
            double a0, a1, a2;
            hipk_mid_poll3<1>(ll, ss_o, ts_o, tt_o, g, seq, fail, ss, a0, a1, a2);
            sbb[tid] = a0;
            sbb[256 + tid] = a1;
            sbb[512 + tid] = a2;
        }
        __syncthreads();
        const double sdot = hipk_mid_tree(sbb, lane), tsd = hipk_mid_tree(sbb + 256, lane), ttd = hipk_mid_tree(sbb + 512, lane);
        buf ^= 1;
        HIPK_BIM_FAIL()
        // ---- K5: omega' -> tests; x += alpha' p (+ omega' s); r = s (- omega' t); <r,r>, <rhat,r>   (TSL:920-961)
        const bool exit_early = sdot < atol2;                                 // TSL:920 (strict)
        const double omega_new = (fabs(ttd) < EPS) ? 0.0 : tsd / ttd;         // TSL:926-930
        if (fabs(omega_new) < EPS && !exit_early) {                           // TSL:934-936
            stop_it = it;
            code = -11;
            extra_mv = 2;
            break;
        }
        const T omn = (T)omega_new;
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int lrow = (t0 + TSTEP * k) * HIPK_TILE + tl;
            const T d_own = PRE ? dw[H + lrow] : (T)1;
            const T p_own = PRE ? d_own * pw[H + lrow] : pw[H + lrow];   // x advances with phat, shat (TSL:942)
            const T s_own = rw[H + lrow], sh_own = PRE ? d_own * s_own : s_own;
            T r_new;
            if (exit_early) {  // TSL:942-950 with exit_early true
                const T m0 = al * p_own;
                xo[k] = xo[k] + m0;
                r_new = s_own;
            } else {
                const T m0 = al * p_own;
                const T m1 = omn * sh_own;
                const T m2 = m0 + m1;
                xo[k] = xo[k] + m2;
                const T m3 = omn * to[k];
                r_new = s_own - m3;
            }
            rw[H + lrow] = r_new;
            if (base + lrow < n && pub[k]) hipk_ll_put(ll, (unsigned)(base + lrow), (double)r_new, seq, r_off);
        }
        __syncthreads();
        sbb = sb + buf * 3 * 256;
        if (tid < 256) {
            double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
            for (int j = 0; j < CH / (256 * VEC); ++j)
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    const int i = VEC * tid + 256 * VEC * j + k;
                    const double rv = (double)rw[H + i], hv = (double)hw[i];
                    if (base + i < n) {
                        acc0 = fma(rv, rv, acc0);
                        acc1 = fma(hv, rv, acc1);
                    }
                }
            sbb[tid] = acc0;
            sbb[256 + tid] = acc1;
        }
        __syncthreads();
        if (tid < 128) {   // wavefront 0: <r,r> of the chunk; 1: <rhat,r>
            const double part = hipk_mid_tree(sbb + 256 * wave, lane);
            if (lane == 0) hipk_ll_put(ll, wg * ss, part, seq, wave == 0 ? rr_o : rhr_o);
            if (wave == 0) part_rr_own = part;
            else part_rhr_own = part;
        }
        buf ^= 1;
        have_parts = true;
        rho = rho_new;
        alpha = alpha_new;
        omega = omega_new;
        ++it;
        iters = it;
        if (exit_early || it >= a.maxiter) {  // TSL:961, loop bound :892
            stop_it = it;
            break;
        }
        // hand-off 3: r at the window's halo columns (the partials are polled by K1 of the next iteration)
        for (int idx = tid; idx < WINc - CH; idx += NTHR) {   // the window without the own tiles (contiguous at H)
            const int widx = idx < H ? idx : idx + CH;
            const int64_t gc = (int64_t)stile[widx >> 8] * HIPK_TILE + (widx & (HIPK_TILE - 1));
            if (gc < n) {
                double v = 0.0;
                if (!hipk_ll_wait(ll, (unsigned)gc, seq, hipk_ll_load(ll, (unsigned)gc, r_off), v, r_off)) *fail = 1;
                rw[widx] = (T)v;
            }
        }
    }
#undef HIPK_BIM_FAIL
    // the state the launch sequence (or the next launch) continues from: x, r, p, q of the own rows; this chunk's partials
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int lrow = (t0 + TSTEP * k) * HIPK_TILE + tl;
        if (base + lrow < n) {
            ((T *)a.x)[base + lrow] = xo[k];
            ((T *)a.r)[base + lrow] = rw[H + lrow];   // (after an omega breakdown this is s: the solve has ended, r is not an output)
            ((T *)a.p)[base + lrow] = pw[H + lrow];
            ((T *)a.q)[base + lrow] = qw[H + lrow];
        }
    }
    if (have_parts && lane == 0) {
        if (wave == 0) a.part_rr[wg] = part_rr_own;
        if (wave == 1) a.part_rhr[wg] = part_rhr_own;
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
#endif  // HIPK_BI_MID_H
