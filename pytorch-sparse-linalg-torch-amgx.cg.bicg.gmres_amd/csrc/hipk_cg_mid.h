// hipk_cg_mid.h -- CG for launch-bound MID-SIZE systems (33 .. 512 reduction chunks, 65 k < n <= 1 M rows, fp64, banded rows of
// <= 12 entries): THE WHOLE LOOP IN ONE LAUNCH, one workgroup per reduction chunk.
//
// A dependent launch costs ~4.5 us end to end on MI355X, so two or three launches per iteration leave these systems at 12-25 us
// per iteration whatever the kernels move.  Here every workgroup keeps ITS chunk of the solve on chip for the whole loop:
//   * x and r of its 2048 rows in registers (thread t owns rows base + 256 tt + t, tt = 0..7: the SpMV's tile layout, so the
//     <p,Ap> tile sums are wavefront sums over 64 contiguous rows as in the SpMV epilogue),
//   * its matrix rows (values + window-relative column indices) in registers,
//   * p at every column its rows reference in LDS: a window [base - H, base + 2048 + H) of the vector, H = the matrix's reach
//     beyond a chunk (hipk_mid_reach_kernel, once per handle).  After <r,r> the workgroup advances the WHOLE window,
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
#include "hipk_handoff.h"

static constexpr int kMidMaxChunks = 512;     // two partials per thread in the fold
static constexpr int kMidSpinBound = 1 << 18; // polls (~1 us each) before a workgroup gives up on a hand-off

__device__ __forceinline__ void hipk_ll_put(unsigned long long *slot, double v, unsigned seq) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v), s = (unsigned long long)seq << 32;
    __hip_atomic_store(slot, (b & 0xffffffffull) | s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(slot + 1, (b >> 32) | s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool hipk_ll_get(const unsigned long long *slot, unsigned seq, double &v) {
    const unsigned long long w0 = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long w1 = __hip_atomic_load(slot + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    v = __longlong_as_double((long long)((w0 & 0xffffffffull) | (w1 << 32)));
    return (unsigned)(w0 >> 32) == seq && (unsigned)(w1 >> 32) == seq;
}
// poll one flagged word; false when the spin bound was hit
__device__ __forceinline__ bool hipk_ll_wait(const unsigned long long *slot, unsigned seq, double &v) {
    unsigned spins = 0;
    while (!hipk_ll_get(slot, seq, v)) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (unsigned)kMidSpinBound) return false;
    }
    return true;
}

// reach of the matrix beyond the reduction chunk of each row: max over entries of the distance of the column from [lo, lo + ch)
__global__ __launch_bounds__(256) void hipk_mid_reach_kernel(const int *__restrict__ crow, const int *__restrict__ col, int64_t n,
                                                             int ch, int *__restrict__ out) {
    int m = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t lo = (i / ch) * ch, hi = lo + ch - 1;
        for (int e = crow[i]; e < crow[i + 1]; ++e) {
            const int64_t cc = col[e];
            const int64_t d = cc < lo ? lo - cc : (cc > hi ? cc - hi : 0);
            m = d > m ? (int)d : m;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const int o = __shfl_xor(m, off);
        m = o > m ? o : m;
    }
    if ((threadIdx.x & 63) == 0 && m > 0) atomicMax(out, m);
}

struct hipk_cg_mid_args {
    int64_t n;
    int g, H;                     // chunks; window reach (a multiple of 128)
    const int *crow, *col;
    const double *val;
    double *x, *r, *p;
    unsigned long long *r_ll;     // [2 n] flagged words of r
    unsigned long long *pap_ll;   // [2 g] chunk partials of <p,Ap>
    unsigned long long *rr_ll;    // [2 g] chunk partials of <r,r>
    hipk_lds_ctl *ctl;
    double *gamma;
    const double *atol2;
    int64_t *stop_it;
    int64_t it0, maxiter, max_its;
    int test_not_resident;
};
static inline size_t hipk_cg_mid_lds_bytes(int H) { return (size_t)(2 * (HIPK_BASE_CHUNK + 2 * H) + 8 + HIPK_THREADS + 32 + 8) * sizeof(double); }

// fold of the G flagged chunk partials in the spec's order (hipk_reduce_parts); *fail set when a partial never arrived
__device__ __forceinline__ double hipk_mid_fold(const unsigned long long *ll, int g, unsigned seq, double *sbuf, int *fail) {
    const int t = threadIdx.x;
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < kMidMaxChunks / HIPK_THREADS; ++k) {
        const int i = t + k * HIPK_THREADS;
        if (i < g) {
            double v;
            if (!hipk_ll_wait(ll + 2 * i, seq, v)) *fail = 1;
            acc = acc + v;
        }
    }
    return hipk_block_sum(acc, sbuf);
}

template <int W>
__global__ __launch_bounds__(HIPK_THREADS) void hipk_cg_mid_kernel(hipk_cg_mid_args a) {
    constexpr int CH = HIPK_BASE_CHUNK, NT = CH / HIPK_TILE;   // 2048 rows = 8 tiles of 256
    extern __shared__ double mid_lds[];
    const int c = blockIdx.x, g = a.g, H = a.H, WIN = CH + 2 * H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double *pw = mid_lds;            // p at the window's columns; pw[WIN] = 0.0 for the padding entries of short rows
    double *rw = pw + WIN + 8;       // r at the window's columns; [H, H + CH) is the own chunk
    double *sbuf = rw + WIN;         // [256]
    double *ts = sbuf + HIPK_THREADS;   // [32] wavefront sums of <p,Ap>, 4 per tile
    int *fail = (int *)(ts + 32);
    const int64_t n = a.n, base = (int64_t)c * CH, w0 = base - H;
    const int ntiles = (int)((n + HIPK_TILE - 1) / HIPK_TILE);
    hipk_lds_ctl *scal = a.ctl;
    if (tid == 0) *fail = 0;

    // ---- the chunk's rows: x, r, matrix entries in registers; the p window in LDS
    double xo[NT], ro[NT], vj[NT][W];
    int cj[NT][W];
    // short rows are padded with (value +0.0, column slot WIN holding +0.0): acc + (0.0 * 0.0) leaves every acc as it is (acc
    // starts at +0.0 and can never become -0.0), so the sum has the bits of the row's own entries added in CSR order -- and no
    // per-entry predicate has to be kept alive over the loop (as 64-bit lane masks they cost 2 W scalar registers per row)
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) {
        const int64_t row = base + tt * HIPK_TILE + tid;
        const bool live = row < n;
        xo[tt] = live ? a.x[row] : 0.0;
        ro[tt] = live ? a.r[row] : 0.0;
        int lo = 0, len = 0;
        if (live) {
            lo = a.crow[row];
            len = a.crow[row + 1] - lo;
        }
#pragma unroll
        for (int j = 0; j < W; ++j) {
            const bool has = j < len;
            cj[tt][j] = has ? (int)(a.col[lo + j] - w0) : WIN;
            vj[tt][j] = has ? a.val[lo + j] : 0.0;
        }
    }
    if (tid < 8) pw[WIN + tid] = 0.0;
    for (int idx = tid; idx < WIN; idx += HIPK_THREADS) {
        const int64_t gc = w0 + idx;
        pw[idx] = (gc >= 0 && gc < n) ? a.p[gc] : 0.0;
        rw[idx] = 0.0;
    }
    double gamma = a.gamma[a.it0 & 1];
    const double atol2 = *a.atol2;
    const int64_t stop0 = *a.stop_it;
    double rs_last = scal->rs_last;

    // every workgroup resident?  Nothing has been modified yet: a failure leaves the solve to the launch sequences
    int epoch = 0;
    if (!hipk_gbar(&scal->bar, g, epoch, fail) || a.test_not_resident) {
        if (tid == 0) scal->redo = -1;
        return;
    }
    unsigned seq = 0;
    int64_t it = a.it0;
    bool done = stop0 <= it;
    while (!done) {
        ++seq;
        // ---- A p of the own rows (products rounded, added in CSR order), wavefront sums of p .* (A p)   (TSL:845-846)
        double Ap[NT];
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < W; ++j) {
                const double pr = vj[tt][j] * pw[cj[tt][j]];
                acc = acc + pr;
            }
            const bool live = base + tt * HIPK_TILE + tid < n;
            Ap[tt] = acc;   // rows beyond n: padding only, +0.0
            double d0 = live ? pw[H + tt * HIPK_TILE + tid] * Ap[tt] : 0.0;
            d0 = hipk_wave_sum(d0);
            if (lane == 0) ts[tt * 4 + wave] = d0;
        }
        __syncthreads();
        if (tid == 0) {   // the chunk's partial of the tiled dot (hipk_tile_combine_kernel's fold)
            double tp[NT];
#pragma unroll
            for (int tt = 0; tt < NT; ++tt)
                tp[tt] = (c * NT + tt < ntiles) ? 0.0 + ((ts[tt * 4] + ts[tt * 4 + 1]) + (ts[tt * 4 + 2] + ts[tt * 4 + 3])) : 0.0;
            const double part = 0.0 + (((tp[0] + tp[4]) + (tp[2] + tp[6])) + ((tp[1] + tp[5]) + (tp[3] + tp[7])));
            hipk_ll_put(a.pap_ll + 2 * c, part, seq);
        }
        const double pAp = hipk_mid_fold(a.pap_ll, g, seq, sbuf, fail);
        if (*fail) {
            if (tid == 0) scal->redo = -3;
            return;
        }
        // ---- alpha, r, x; r to the neighbours; the chunk's partial of <r,r>   (TSL:846-850)
        const double alpha = gamma / pAp;
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
            const int64_t row = base + tt * HIPK_TILE + tid;
            const double m1 = alpha * Ap[tt];
            ro[tt] = ro[tt] - m1;
            const double m0 = alpha * pw[H + tt * HIPK_TILE + tid];
            xo[tt] = xo[tt] + m0;
            rw[H + tt * HIPK_TILE + tid] = ro[tt];
            if (row < n) hipk_ll_put(a.r_ll + 2 * row, ro[tt], seq);
        }
        __syncthreads();
        {
            double acc = 0.0;   // virtual thread t of the chunk: elements {2t, 2t+1} + 512 j ascending (the plain dot of the spec)
#pragma unroll
            for (int j = 0; j < CH / (2 * HIPK_THREADS); ++j)
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int i = 2 * tid + 2 * HIPK_THREADS * j + k;
                    const double v = rw[H + i];
                    if (base + i < n) acc = fma(v, v, acc);
                }
            acc = hipk_block_sum(acc, sbuf);
            if (tid == 0) hipk_ll_put(a.rr_ll + 2 * c, acc, seq);
        }
        // r at the window's halo columns (issued before the fold of <r,r>: the neighbours published r before their partial)
        for (int idx = tid; idx < 2 * H; idx += HIPK_THREADS) {
            const int widx = idx < H ? idx : idx + CH;
            const int64_t gc = w0 + widx;
            if (gc >= 0 && gc < n) {
                double v;
                if (!hipk_ll_wait(a.r_ll + 2 * gc, seq, v)) *fail = 1;
                rw[widx] = v;
            }
        }
        const double rr = hipk_mid_fold(a.rr_ll, g, seq, sbuf, fail);
        if (*fail) {
            if (tid == 0) scal->redo = -3;
            return;
        }
        // ---- beta, p over the whole window, stop test   (TSL:851-853, 841)
        const double beta = rr / gamma;
        for (int idx = tid; idx < WIN; idx += HIPK_THREADS) {
            const double m = beta * pw[idx];
            pw[idx] = rw[idx] + m;
        }
        __syncthreads();
        gamma = rr;
        rs_last = rr;
        ++it;
        done = (it >= a.maxiter || rr <= atol2);
        if (it - a.it0 >= a.max_its) break;
    }
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) {
        const int64_t row = base + tt * HIPK_TILE + tid;
        if (row < n) {
            a.x[row] = xo[tt];
            a.r[row] = ro[tt];
            a.p[row] = pw[H + tt * HIPK_TILE + tid];
        }
    }
    if (c == 0 && tid == 0) {
        a.gamma[it & 1] = gamma;
        scal->rs_last = rs_last;
        scal->it_done = it;
        if (done && it < stop0) *a.stop_it = it;
    }
}
#endif  // HIPK_CG_MID_H
