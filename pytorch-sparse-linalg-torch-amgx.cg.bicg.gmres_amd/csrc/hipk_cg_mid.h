// hipk_cg_mid.h -- CG for launch-bound MID-SIZE systems (33 .. 512 reduction chunks, 65 k < n <= 1 M rows, fp64, banded rows of
// <= 12 entries): THE WHOLE LOOP IN ONE LAUNCH, one workgroup per reduction chunk.
//
// A dependent launch costs ~4.5 us end to end on MI355X, so two or three launches per iteration leave these systems at 12-25 us
// per iteration whatever the kernels move.  Here every workgroup keeps ITS chunk of the solve on chip for the whole loop:
//   * x and r of its 2048 rows in registers (thread t owns rows base + 256 tt + t, tt = 0..7: the SpMV's tile layout, so the
//     <p,Ap> tile sums are wavefront sums over 64 contiguous rows as in the SpMV epilogue),
//   * its matrix rows (values + window-relative column indices) in registers,
//   * p at every column its rows reference in LDS: a window of 256-column tiles -- the own ones and every tile the rows reference
//     (hipk_mid_plan_get, once per handle; contiguous for a 2-D stencil, three bands for a 3-D one).  After <r,r> the workgroup advances the WHOLE window,
//     p_j = r_j + beta p_j -- the owner's formula on the owner's operands, the same bits -- so p itself is never exchanged
//     (the trick of hipk_cg_solve_lds_kernel / hipk_cg2_spmv_kernel).
// What crosses workgroups per iteration: one chunk partial of <p,Ap>, one of <r,r>, and r (for the neighbours' windows).  All
// three travel as "flagged words" (the LL scheme of collective libraries): a double is stored as two 8-byte words, each carrying
// 32 payload bits and the 32-bit hand-off number, by relaxed agent-scope stores; the consumer polls the PAYLOAD until both
// numbers match.  No separate flag, no store drain, no barrier: a hand-off is one store trip + one load trip through the fabric
// (hipk_ho_sync: drain + flag store + flag poll + payload load = four).  Every workgroup folds all G partials itself in the
// spec's order (thread t takes partial t, t + 256; tree 128 .. 1), so all of them derive the same alpha / beta bits.
// Slot reuse is safe without double buffering: a producer overwrites hand-off k's word with k + 1 only after it has folded a
// partial that EVERY consumer publishes after having consumed k (pAp(k+1) <- all windows advanced <- r(k), <r,r>(k) consumed).
// Arithmetic per element = the launch sequence's (TSL:845-853), bit for bit; tests/test_gpu_api.py::test_cg_mid_one_launch_*.
#ifndef HIPK_CG_MID_H
#define HIPK_CG_MID_H
#include "hipk_mid.h"

struct hipk_cg_mid_args {
    int64_t n;
    int g, win;                   // chunks; doubles of a window in LDS = 256 x the most tiles any workgroup's window holds
    hipk_mid_plan plan;           // which tiles (hipk_mid.h)
    const int *crow, *col;
    const void *val;              // values, x, r, p, dinv: of the handle's dtype (the kernel's T)
    void *x, *r, *p;
    unsigned long long *r_ll;     // [2 n] flagged words of r
    unsigned long long *pap_ll;   // [2 g] chunk partials of <p,Ap>
    unsigned long long *rr_ll;    // [2 g] chunk partials of <r,r>
    unsigned long long *rz_ll;    // PRE: chunk partials of <r, M r>
    const void *dinv;             // PRE: the Jacobi preconditioner's diagonal, M = diag(dinv) (TSL:849)
    const double *rz0_parts;      // PRE, it0 = 0: chunk partials of gamma0 = <r0, M r0> (hipk_pcg_start_kernel)
    hipk_lds_ctl *ctl;
    double *gamma;
    const double *atol2;
    int64_t *stop_it;
    int64_t it0, maxiter, max_its;
    int test_not_resident;
    int slot_stride;              // distance of consecutive chunk-partial slots in 16-byte words (16: a 256-byte line each)
    int xcd_aware;                // 0 (HIPK_CG_MID_XCD=0, A/B measurements): workgroup b takes row range b
};
// LDS of a workgroup that owns `nch` chunks, windows of `win` doubles: p window + 8 zero slots | r window | 2 fold buffers | tile sums |
// flag | (pre: the window of dinv; two sums per fold) | the window's tile list
static inline size_t hipk_cg_mid_lds_bytes(int win, int nch, bool pre = false, size_t sv = 8) {   // sv: bytes of a vector element
    return (size_t)((pre ? 3 : 2) * win + 8) * sv + (size_t)(2 * 256 * nch * (pre ? 2 : 1) + 32 * nch + 8 + kMidPlanSlots / 2) * sizeof(double);
}

// Diagnostic twin (make stamps): thread 0 of every workgroup sums, over the iterations of a launch, the constant 100 MHz clock
// between its phase boundaries; tools/cg_mid_stamps_probe.py prints where an iteration goes.
#ifdef HIPK_GM_STAMPS
#define HIPK_MID_NSTAMP 12
__device__ unsigned long long hipk_mid_stamps[kMidMaxChunks * HIPK_MID_NSTAMP];
#define HIPK_MSTAMP(k)                                                  \
    do {                                                                \
        const unsigned long long t_ = __builtin_amdgcn_s_memrealtime(); \
        st_acc[k] += t_ - st_prev;                                      \
        st_prev = t_;                                                   \
    } while (0)
#else
#define HIPK_MSTAMP(k)
#endif

// W: matrix entries per row held in registers; NCH: reduction chunks per workgroup (1: up to n_cu chunks; 2: beyond -- half the
// rows then sit inside a workgroup's own window and are not published).  1024 threads, 2 NCH rows each, one workgroup per CU.
// PRE (NCH = 1 only): Jacobi-preconditioned CG -- z = dinv .* r is formed where it is used (never stored), gamma = <r,z> steers
// alpha and beta, <r,r> the stop test (TSL:835-841, 849-852; hipk_pcg_update_kernel / hipk_pcg_direction_kernel bit for bit).
// T: the handle's dtype -- vectors, windows and the element-wise arithmetic in T (fp32 storage: the launch kernels' extension), dots
// and scalars in double as everywhere.
template <typename T, int W, int NCH, bool PRE = false>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4, 4))) void hipk_cg_mid_kernel(hipk_cg_mid_args a) {
    static_assert(!PRE || NCH == 1, "the preconditioned form owns one chunk per workgroup");
    constexpr int NSB = 256 * NCH * (PRE ? 2 : 1);   // doubles of one fold buffer
    constexpr int NTHR = 1024, CH = HIPK_BASE_CHUNK, NT = CH / HIPK_TILE;   // a chunk: 2048 rows = 8 tiles of 256
    constexpr int OWN = NCH * CH, R = OWN / NTHR, TSTEP = NTHR / HIPK_TILE;  // rows per thread; tiles one pass of the workgroup covers
    extern __shared__ double mid_lds[];
    // XCD-aware placement (blocks b and b + 8 share an XCD): XCD k takes the k-th contiguous eighth of the workgroups' row
    // ranges, so that a window's neighbours mostly sit behind the same L2 -- their write-through r then HITS there (a 0.1 us
    // poll instead of a 0.5 us trip through the fabric).  Speed only; the grid is padded to a multiple of 8, idle blocks leave.
    const int g = a.g, WIN = a.win, nwg = (g + NCH - 1) / NCH;
    const int wg = a.xcd_aware ? hipk_xcd_chunk(blockIdx.x, nwg) : ((int)blockIdx.x < nwg ? (int)blockIdx.x : -1);
    if (wg < 0) return;
    const int tid = threadIdx.x, lane = tid & 63, tw = (tid >> 6) & 3, tl = tid & (HIPK_TILE - 1), t0 = tid >> 8;
    T *pw = (T *)mid_lds;            // p at the window's columns; pw[WIN] = 0.0 for the padding entries of short rows
    T *rw = pw + WIN + 8;       // r at the window's columns; [H, H + OWN) are the own rows (H: where the own tiles sit in the window)
    double *sb = (double *)(rw + WIN);   // 2 x [NSB]: fold buffers, used alternately (one barrier per fold)
    double *ts = sb + 2 * NSB;       // [32 NCH] wavefront sums of <p,Ap>, 4 per tile
    int *fail = (int *)(ts + 32 * NCH);
    T *dw = (T *)(fail + 2);         // PRE: dinv at the window's columns
    int *stile = (int *)(dw + (PRE ? WIN : 0));   // the window's tiles: slot s holds columns 256 stile[s] .. + 255
    const int64_t n = a.n, base = (int64_t)wg * OWN;
    const int tlo = a.plan.tlo[wg], WINc = a.plan.nslot[wg] * HIPK_TILE;   // this workgroup's window
    const short *tmap = a.plan.map + (size_t)wg * kMidPlanRange;
    const int H = __builtin_amdgcn_readfirstlane((int)tmap[(int)(base >> 8) - tlo] * HIPK_TILE);   // (uniform: a 16-bit load lands in a VGPR)
    if (tid < kMidPlanSlots) stile[tid] = (tid * HIPK_TILE < WINc) ? a.plan.tiles[wg * kMidPlanSlots + tid] : 0;
    __syncthreads();
    const int ntiles = (int)((n + HIPK_TILE - 1) / HIPK_TILE);
    hipk_lds_ctl *scal = a.ctl;
    if (tid == 0) *fail = 0;
    const hipk_ll_rsrc r_ll = hipk_ll_make(a.r_ll, (size_t)n * 16), pap_ll = hipk_ll_make(a.pap_ll, (size_t)g * a.slot_stride * 16),
                       rr_ll = hipk_ll_make(a.rr_ll, (size_t)g * a.slot_stride * 16),
                       rz_ll = hipk_ll_make(PRE ? a.rz_ll : a.rr_ll, (size_t)g * a.slot_stride * 16);
    const int ss = a.slot_stride;

    // ---- the own rows: x, r, matrix entries in registers (thread t: rows 256 (t0 + 4 k) + tl); the p window in LDS
    T xo[R], ro[R], vj[R][W];
    int cj[R][W];
    // short rows are padded with (value +0.0, column slot WIN holding +0.0): acc + (0.0 * 0.0) leaves every acc as it is (acc
    // starts at +0.0 and can never become -0.0), so the sum has the bits of the row's own entries added in CSR order -- and no
    // per-entry predicate has to be kept alive over the loop (as 64-bit lane masks they cost 2 W scalar registers per row)
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int64_t row = base + (t0 + TSTEP * k) * HIPK_TILE + tl;
        const bool live = row < n;
        xo[k] = live ? ((const T *)a.x)[row] : (T)0;
        ro[k] = live ? ((const T *)a.r)[row] : (T)0;
        int lo = 0, len = 0;
        if (live) {
            lo = a.crow[row];
            len = a.crow[row + 1] - lo;
        }
#pragma unroll
        for (int j = 0; j < W; ++j) {
            const bool has = j < len;
            const int cc = has ? a.col[lo + j] : 0;
            cj[k][j] = has ? (int)tmap[(cc >> 8) - tlo] * HIPK_TILE + (cc & (HIPK_TILE - 1)) : WIN;
            vj[k][j] = has ? ((const T *)a.val)[lo + j] : (T)0;
        }
    }
    if (tid < 8) pw[WIN + tid] = (T)0;
    for (int idx = tid; idx < WINc; idx += NTHR) {
        const int64_t gc = (int64_t)stile[idx >> 8] * HIPK_TILE + (idx & (HIPK_TILE - 1));
        pw[idx] = gc < n ? ((const T *)a.p)[gc] : (T)0;
        rw[idx] = (T)0;
        if (PRE) dw[idx] = gc < n ? ((const T *)a.dinv)[gc] : (T)0;
    }
    double gamma = a.gamma[a.it0 & 1];
    const double atol2 = *a.atol2;
    const int64_t stop0 = *a.stop_it;
    double rs_last = scal->rs_last;
    // rows whose r another workgroup's window holds (their tile is in that window's list)
    bool pub[R];
#pragma unroll
    for (int k = 0; k < R; ++k) pub[k] = a.plan.needed[(int)(base >> 8) + t0 + TSTEP * k] != 0;

    // every workgroup resident?  Nothing has been modified yet: a failure leaves the solve to the launch sequences
    int epoch = 0;
    if (!hipk_gbar(&scal->bar, nwg, epoch, fail) || a.test_not_resident) {
        if (tid == 0) scal->redo = -1;
        return;
    }
    if (PRE && a.it0 == 0) {   // gamma0 = <r0, M r0>: chunk partials of the launch before this one (hipk_reduce_parts)
        if (tid < 256) {
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < kMidMaxChunks / 256; ++k)
                if (tid + k * 256 < g) acc = acc + a.rz0_parts[tid + k * 256];
            sb[tid] = acc;
        }
        __syncthreads();
        gamma = hipk_mid_tree(sb, lane);
        __syncthreads();
    }
    unsigned seq = 0;
    int buf = 0;
    int64_t it = a.it0;
    bool done = stop0 <= it;
#ifdef HIPK_GM_STAMPS
    unsigned long long st_acc[HIPK_MID_NSTAMP] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_prev = __builtin_amdgcn_s_memrealtime();
#endif
    const int H_ = H, WIN_ = WIN;
    while (!done) {
        ++seq;
        // the window geometry is made opaque once per iteration: otherwise every LDS address of the loop is formed ahead of it and
        // kept in a register (hipk_bi_mid.h: 25 spilled VGPRs became 1)
        int H = H_, WIN = WIN_;
        asm volatile("" : "+s"(H), "+s"(WIN));
        T *rw = pw + WIN + 8;
        T *dw = (T *)((int *)((double *)(rw + WIN) + 2 * NSB + 32 * NCH) + 2);
        // ---- A p of the own rows (products rounded, added in CSR order), wavefront sums of p .* (A p)   (TSL:845-846)
        T Ap[R];
#pragma unroll
        for (int k = 0; k < R; k += 2) {
            double d[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int lrow = (t0 + TSTEP * (k + h)) * HIPK_TILE + tl;
                T acc = (T)0;
#pragma unroll
                for (int j = 0; j < W; ++j) {
                    const T pr = vj[k + h][j] * pw[cj[k + h][j]];
                    acc = acc + pr;
                }
                Ap[k + h] = acc;   // rows beyond n: padding only, +0.0
                if (R > 2) rw[H + lrow] = acc;   // four rows per thread: A p waits in the (idle) own slot of the r window
                d[h] = (base + lrow < n) ? (double)pw[H + lrow] * (double)acc : 0.0;
            }
            const double s2 = hipk_wave_sum_pair(d[0], d[1]);
            if ((lane & 31) == 0) ts[(t0 + TSTEP * (k + (lane >> 5))) * 4 + tw] = s2;
        }
        __syncthreads();
        HIPK_MSTAMP(0);
        if (tid < 64) {   // the chunks' partials of the tiled dot (hipk_tile_combine_kernel's fold): lanes 8 q .. 8 q + 7 a tile each
            const int q = lane >> 3;
            double tp = 0.0;
            if (lane < NT * NCH && (wg * NCH) * NT + lane < ntiles) tp = 0.0 + ((ts[lane * 4] + ts[lane * 4 + 1]) + (ts[lane * 4 + 2] + ts[lane * 4 + 3]));
            tp = tp + hipk_row_shl<4>(tp);   // (p0+p4) (p1+p5) (p2+p6) (p3+p7)
            tp = tp + hipk_row_shl<2>(tp);   // (p0+p4)+(p2+p6)  (p1+p5)+(p3+p7)
            tp = tp + hipk_row_shl<1>(tp);
            if ((lane & 7) == 0 && q < NCH && wg * NCH + q < g) hipk_ll_put(pap_ll, (wg * NCH + q) * ss, 0.0 + tp, seq);
        }
        HIPK_MSTAMP(1);
        if (tid < 256) sb[buf * NSB + tid] = hipk_mid_poll<kMidMaxChunks / 256, true>(pap_ll, g, seq, fail, ss);
        HIPK_MSTAMP(2);
        __syncthreads();
        const double pAp = hipk_mid_tree(sb + buf * NSB, lane);
        buf ^= 1;
        HIPK_MSTAMP(3);
        if (*fail) {
            if (tid == 0) scal->redo = -3;
            return;
        }
        // ---- alpha, r, x; r to the neighbours; the chunks' partials of <r,r>   (TSL:846-850)
        const T alpha = (T)(gamma / pAp);
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int lrow = (t0 + TSTEP * k) * HIPK_TILE + tl;
            const T Ap_k = (R > 2) ? rw[H + lrow] : Ap[k];
            const T m1 = alpha * Ap_k;
            ro[k] = ro[k] - m1;
            const T m0 = alpha * pw[H + lrow];
            xo[k] = xo[k] + m0;
            rw[H + lrow] = ro[k];
            if (base + lrow < n && pub[k]) hipk_ll_put(r_ll, (unsigned)(base + lrow), (double)ro[k], seq);   // (a float travels as the double it equals)
        }
        __syncthreads();
        HIPK_MSTAMP(4);
        if (tid < 256 * NCH) {
            const int q = tid >> 8, t = tid & 255;
            constexpr int VEC = hipk_vec<T>::VEC;
            double acc = 0.0, acc1 = 0.0;   // virtual thread t of chunk q: elements {VEC t .. VEC t + VEC - 1} + 256 VEC j ascending (the plain dot of the spec)
#pragma unroll
            for (int j = 0; j < CH / (256 * VEC); ++j)
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    const int i = q * CH + VEC * t + 256 * VEC * j + k;
                    const T rv = rw[H + i];
                    const double v = (double)rv;
                    if (base + i < n) acc = fma(v, v, acc);
                    if (PRE) {
                        const T z = dw[H + i] * rv;   // TSL:849
                        if (base + i < n) acc1 = fma(v, (double)z, acc1);   // TSL:850
                    }
                }
            sb[buf * NSB + tid] = acc;
            if (PRE) sb[buf * NSB + 256 + tid] = acc1;
        }
        HIPK_MSTAMP(5);
        __syncthreads();
        if (tid < 64 * NCH) {   // wavefront q folds chunk q's 256 chains and publishes the partial
            const int q = tid >> 6;
            const double part = hipk_mid_tree(sb + buf * NSB + q * 256, lane);
            if (lane == 0 && wg * NCH + q < g) hipk_ll_put(rr_ll, (wg * NCH + q) * ss, part, seq);
        } else if (PRE && tid < 128) {
            const double part = hipk_mid_tree(sb + buf * NSB + 256, lane);
            if (lane == 0) hipk_ll_put(rz_ll, wg * ss, part, seq);
        }
        buf ^= 1;
        HIPK_MSTAMP(6);
        // r at the window's halo columns (polled before the partials of <r,r>: the neighbours published r before theirs).
        // (Issuing these loads earlier and examining them after the fold measured slower: a poll is a 0.5 us trip, the early
        // ones mostly came back empty and the partials' polls queued behind them -- 7.1 vs 5.8 us per iteration at n = 250 k.)
        for (int idx = tid; idx < WINc - OWN; idx += NTHR) {   // the window without the own tiles (contiguous at H)
            const int widx = idx < H ? idx : idx + OWN;
            const int64_t gc = (int64_t)stile[widx >> 8] * HIPK_TILE + (widx & (HIPK_TILE - 1));
            if (gc < n) {
                double v = 0.0;
                if (!hipk_ll_wait<true>(r_ll, (unsigned)gc, seq, hipk_ll_load(r_ll, (unsigned)gc), v)) *fail = 1;
                rw[widx] = (T)v;
            }
        }
        HIPK_MSTAMP(7);
        if (tid < 256) {
            sb[buf * NSB + tid] = hipk_mid_poll<kMidMaxChunks / 256, true>(rr_ll, g, seq, fail, ss);
            if (PRE) sb[buf * NSB + 256 + tid] = hipk_mid_poll<kMidMaxChunks / 256, true>(rz_ll, g, seq, fail, ss);
        }
        HIPK_MSTAMP(8);
        __syncthreads();
        const double rr = hipk_mid_tree(sb + buf * NSB, lane);
        const double gamma_new = PRE ? hipk_mid_tree(sb + buf * NSB + 256, lane) : rr;   // <r,z> steers alpha and beta, <r,r> the stop test
        buf ^= 1;
        HIPK_MSTAMP(9);
        if (*fail) {
            if (tid == 0) scal->redo = -3;
            return;
        }
        // ---- beta, p over the whole window, stop test   (TSL:851-853, 841)
        const T beta = (T)(gamma_new / gamma);
        for (int idx = tid; idx < WINc; idx += NTHR) {
            const T zj = PRE ? dw[idx] * rw[idx] : rw[idx];   // z = M r (TSL:849), formed again: same operands, same bits
            const T m = beta * pw[idx];
            pw[idx] = zj + m;
        }
        __syncthreads();
        HIPK_MSTAMP(10);
        gamma = gamma_new;
        rs_last = rr;
        ++it;
        done = (it >= a.maxiter || rr <= atol2);
        if (it - a.it0 >= a.max_its) break;
    }
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int lrow = (t0 + TSTEP * k) * HIPK_TILE + tl;
        if (base + lrow < n) {
            ((T *)a.x)[base + lrow] = xo[k];
            ((T *)a.r)[base + lrow] = ro[k];
            ((T *)a.p)[base + lrow] = pw[H + lrow];
        }
    }
#ifdef HIPK_GM_STAMPS
    if (tid == 0)
        for (int k = 0; k < HIPK_MID_NSTAMP; ++k) hipk_mid_stamps[wg * HIPK_MID_NSTAMP + k] = st_acc[k];
#endif
    if (wg == 0 && tid == 0) {
        a.gamma[it & 1] = gamma;
        scal->rs_last = rs_last;
        scal->it_done = it;
        if (done && it < stop0) *a.stop_it = it;
    }
}
#endif  // HIPK_CG_MID_H
