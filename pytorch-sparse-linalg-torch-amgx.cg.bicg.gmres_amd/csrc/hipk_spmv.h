// hipk_spmv.h -- CSR SpMV for gfx950: lean 256-row tiles, LDS-staged products, fused dots.
//
// Replaces `torch.matmul(A, v)` (TSL:191) and, fused, the dot that follows it in
// every solver loop (TSL:846 <p,Ap>; TSL:910 <rhat,q>; TSL:926-930 <t,s>,<t,t>;
// TSL:352 ||A v||) and the residual form b - A x (TSL:820, 875, 791, 1008).
//
// One workgroup of 256 threads per TILE of 256 consecutive rows (matrix independent):
//   * the tile's entries are read fully coalesced (lane i <-> entry j0+i): `col` with the
//     default cache policy, `val` NON-TEMPORAL -- val is the one pure stream (160 MB at
//     N=4M); keeping it out of the 256 MiB Infinity Cache lets x, y, crow and col
//     (and, inside a solve, the Krylov vectors) stay resident there (profiles/: 65 -> 56 us);
//   * x is gathered (L2/Infinity Cache hits: XCD-aware tile placement), the PRODUCTS go
//     to LDS, thread t then sums row t's products from LDS in CSR order (rows longer than
//     HIPK_LONG_ROW are summed by a whole wavefront, lanes strided, fixed tree);
//   * thread t writes y[t] and forms w_t*y_t / y_t*y_t; the 256 values are folded (shuffle tree per
//     wavefront; the combine kernel forms ((s0+s1)+(s2+s3))) = the TILE partial; a tiny second kernel folds the tile partials of each
//     reduction chunk into the chunk partial the consumers expect ("tiled dot" spec, mirrored
//     by oracle/krylov_oracle.c).  8 workgroups per CU are resident (11 KB LDS, ~40 VGPRs):
//     latency is hidden by occupancy, not by a software pipeline (measured both, profiles/).
//   * tiles with more than CAP entries take a general path: groups of rows whose entries fit
//     CAP, one row longer than CAP streamed by a single wavefront.
// The kernel is HBM/Infinity-Cache bound: 12 B/nnz + 4 B/row of matrix, 8 B/row in, 8 B/row out.
#pragma once
#include "hipk_common.h"

#define HIPK_SPMV_DOT_W 1   // part0[c] = sum w_i * out_i
#define HIPK_SPMV_DOT_YY 2  // part1[c] = sum out_i * out_i
#define HIPK_SPMV_RESID 4   // out = bsub - A x  instead of A x
#define HIPK_SPMV_SCALE 8   // out = dscale .* out, after RESID and before the fused dots: left Jacobi preconditioning
                            // M(A v), M(b - A x) of the preconditioned GMRES (TSL:351, 791); internal, not in hipk_spmv_ex
#define HIPK_LONG_ROW 32    // rows with more nnz are summed by a wavefront (strided + tree)

struct hipk_spmv_args {
    const int *crow;
    const int *col;
    const void *val;
    const void *x;
    void *y;
    int64_t n;
    int ch;
    int g;
    int mode;
    const void *w;
    const void *bsub;
    double *part0;
    double *part1;
    const int64_t *stop_it;  // device word: kernels with it >= *stop_it do nothing (may be null)
    int64_t it;
    double *tpart0;  // tile partials (handle-owned scratch), filled by the launcher
    double *tpart1;
    const int *row_list;  // row-per-wavefront kernel: process only these rows (null: all rows)
    int n_list;
    // coded path (hipk_coded.h): one byte per entry + one byte per row + a dictionary of (col - row, value) pairs
    const unsigned char *code;
    const unsigned char *rowlen;
    const int *dict_off;
    const void *dict_val;
    int n_codes;
    int code_cap;  // bytes of LDS per tile for the code bytes
    const int *tile_off;  // sliced-ELL layout: prefix sum of the tile sizes (units of 256 bytes), ntiles + 1
    int sell_w;           //   > 0: every tile has this size
    const void *sell_vals;  // offset-coded layout: value planes (same tile prefix as the code planes), else null
    const unsigned long long *tile_ucode;  // per tile: the 8 code bytes every row of the tile shares, 0 = rows differ
    const unsigned long long *tile_wcode;  // two-rows-per-lane kernel: tile_ucode, or the union pattern of a MASKED tile (byte 7 =
                                           //   HIPK_SELL_MASKED) whose rows have subsets of it (row_mask), or 0
    const unsigned char *row_mask;         // per row of a masked tile: bit k = entry k of the pattern is present
    const void *dscale;     // HIPK_SPMV_SCALE: row scaling vector
    int group_tiles;        // hipk_spmv_sell_loop_kernel<.., CHUNKED = true>: > 0 = tiles per workgroup on a grid of groups (set by the
                            //   launcher), 0 = a workgroup per reduction chunk
    int skip_combine;       // small systems: leave the fused dots as per-wavefront tile sums (hipk_csr_s::tile_part); the
                            //   consumer folds them itself (hipk_fold_tiles8) -- one launch less per SpMV
};

#ifdef __HIPCC__
#define HIPK_TILE 256

template <typename T>
__device__ __forceinline__ T hipk_ld_nt(const T *p) {
    return __builtin_nontemporal_load(p);
}

// XCD-aware placement of tiles: workgroups b and b+8 share an XCD (round-robin dispatch), so XCD k
// gets the k-th contiguous eighth of the tiles and the x lines a stencil re-reads hit in ITS L2.
__device__ __forceinline__ int hipk_xcd_tile(int b, int ntiles) {
    const int per = (ntiles + 7) >> 3;
    const int c = (b & 7) * per + (b >> 3);
    return (c < ntiles && (b >> 3) < per) ? c : -1;
}

// FAST = true: handle-creation analysis proved that every tile has <= CAP entries and no row more than
// HIPK_LONG_ROW, so the long-row pass and the general path are compiled out (26 VGPRs instead of 46).
template <typename T, int CAP, bool FAST>
__global__ __launch_bounds__(HIPK_THREADS) void hipk_spmv_kernel(hipk_spmv_args a) {
    constexpr int NI = CAP / HIPK_THREADS;
    static_assert(CAP % HIPK_THREADS == 0, "CAP must be a multiple of the workgroup size");
    static_assert(CAP * sizeof(T) >= 512 * sizeof(double), "prod[] doubles as reduction scratch");
    static_assert(HIPK_TILE == HIPK_THREADS, "one row per thread");

    const int ntiles = (int)((a.n + HIPK_TILE - 1) / HIPK_TILE);
    const int tile = hipk_xcd_tile(blockIdx.x, ntiles);
    if (tile < 0) return;

    __shared__ __attribute__((aligned(16))) T prod[CAP];
    __shared__ int crowL[HIPK_TILE + 1];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    const int *__restrict__ crow = a.crow;
    const int *__restrict__ col = a.col;
    const T *__restrict__ val = (const T *)a.val;
    const T *__restrict__ x = (const T *)a.x;
    T *y = (T *)a.y;  // not restrict: the general path reads pre-pass results from y before overwriting them
    const int mode = a.mode;

    const int64_t r0 = (int64_t)tile * HIPK_TILE;
    const int nr = (int)((a.n - r0 < HIPK_TILE) ? (a.n - r0) : HIPK_TILE);
    // operands of the epilogue, requested now so their latency hides behind the tile's main loads
    // (measured: SpMV+dot inside CG 68.3 -> 66.2 us, stand-alone 59.9 -> 57.7 us)
    T wrow = (T)0, brow = (T)0, drow = (T)0;
    if (t < nr) {
        if (mode & HIPK_SPMV_DOT_W) wrow = ((const T *)a.w)[r0 + t];
        if (mode & HIPK_SPMV_RESID) brow = ((const T *)a.bsub)[r0 + t];
        if (mode & HIPK_SPMV_SCALE) drow = ((const T *)a.dscale)[r0 + t];
    }
    int crow_t = 0, crow_e = 0;
    if (t < nr) crow_t = crow[r0 + t];
    if (t == 0) crow_e = crow[r0 + nr];  // nr can be 256: one more pointer than threads
    // the stop word is read with the tile's first loads already in flight; nothing has been stored yet
    if (a.stop_it != nullptr && a.it >= *a.stop_it) return;
    if (t < nr) crowL[t] = crow_t;
    if (t == 0) crowL[nr] = crow_e;
    __syncthreads();
    const int j0 = crowL[0];
    const int cnt = crowL[nr] - j0;
    T yrow = (T)0;  // row t of the tile

    if (FAST || cnt <= CAP) {
        // ---------------- fast path: the whole tile in one shot
        // Branch-free loads: lanes past the tile's last entry re-read entry 0 (a cached line) instead of being
        // predicated off -- a predicate per load had cost a branch and a basic block per load.  Their products land
        // in prod[j >= cnt], which no row reads.  All column indices first, then all values: vmcnt retires in
        // order, so the x gathers can start as soon as the indices are back while the value loads are in flight.
        if (cnt > 0) {  // uniform
            const int jb = __builtin_amdgcn_readfirstlane(j0);
            const int *__restrict__ colb = col + jb;
            const T *__restrict__ valb = val + jb;
            unsigned jj[NI];
            int cc[NI];
            T vv[NI], xv[NI];
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int j = t + i * HIPK_THREADS;
                jj[i] = (unsigned)(j < cnt ? j : 0);
            }
#pragma unroll
            for (int i = 0; i < NI; ++i) cc[i] = colb[jj[i]];
#pragma unroll
            for (int i = 0; i < NI; ++i) vv[i] = hipk_ld_nt(valb + jj[i]);
#pragma unroll
            for (int i = 0; i < NI; ++i) xv[i] = x[cc[i]];
#pragma unroll
            for (int i = 0; i < NI; ++i) prod[t + i * HIPK_THREADS] = vv[i] * xv[i];
        }
        int is_long = 0;
        int lo = 0, len = 0;
        if (t < nr) {
            lo = crowL[t] - j0;
            len = crowL[t + 1] - j0 - lo;
            is_long = (!FAST && len > HIPK_LONG_ROW) ? 1 : 0;
        }
        // one barrier both publishes the products and tells every wave whether a long row exists
        const int any_long = FAST ? (__syncthreads(), 0) : __syncthreads_or(is_long);
        if (t < nr && !is_long) {
            T s = (T)0;
            for (int j = 0; j < len; ++j) s = s + prod[lo + j];
            yrow = s;
        }
        if (!FAST && any_long) {
            // long rows: one wavefront per row, lanes strided by 64, fixed tree; result handed to thread r via LDS
            __shared__ T ylong[HIPK_TILE];
            for (int r = wave; r < nr; r += HIPK_THREADS / 64) {
                const int lo2 = crowL[r] - j0;
                const int hi = crowL[r + 1] - j0;
                if (hi - lo2 > HIPK_LONG_ROW) {
                    T s = (T)0;
                    for (int j = lo2 + lane; j < hi; j += 64) s = s + prod[j];
                    s = hipk_wave_sum(s);
                    if (lane == 0) ylong[r] = s;
                }
            }
            __syncthreads();
            if (is_long) yrow = ylong[t];
        }
    } else if (!FAST) {
        // ---------------- general path: groups of rows whose entries fit CAP
        __shared__ T ystage[HIPK_TILE];
        int ra = 0;
        while (ra < nr) {
            const int g0 = crowL[ra];
            int hi = nr;
            int re;
            if (crowL[hi] - g0 <= CAP) {
                re = hi;
            } else {
                int lo = ra;
                while (hi - lo > 1) {
                    const int mid = (lo + hi) >> 1;
                    if (crowL[mid] - g0 <= CAP) lo = mid; else hi = mid;
                }
                re = lo;
            }
            if (re == ra) {
                // one row with more than CAP entries: its sum was computed by the row-per-wavefront pre-pass
                // (hipk_spmv_rowwave_kernel over hipk_csr_s::huge_rows) and sits in y
                if (t == 0) ystage[ra] = y[r0 + ra];
                __syncthreads();
                ra += 1;
                continue;
            }
            const int gcnt = crowL[re] - g0;
            {  // branch-free, as on the fast path: surplus lanes re-read the group's first entry (gcnt >= 1 here)
                const int gb = __builtin_amdgcn_readfirstlane(g0);
                const int *__restrict__ colb = col + gb;
                const T *__restrict__ valb = val + gb;
                int cc[NI];
                T vv[NI];
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const int j = t + i * HIPK_THREADS;
                    cc[i] = colb[(unsigned)(j < gcnt ? j : 0)];
                }
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const int j = t + i * HIPK_THREADS;
                    vv[i] = hipk_ld_nt(valb + (unsigned)(j < gcnt ? j : 0));
                }
#pragma unroll
                for (int i = 0; i < NI; ++i) prod[t + i * HIPK_THREADS] = vv[i] * x[cc[i]];
            }
            __syncthreads();
            int is_long = 0;
            {
                const int r = ra + t;
                if (r < re) {
                    const int lo = crowL[r] - g0;
                    const int len = crowL[r + 1] - g0 - lo;
                    if (len <= HIPK_LONG_ROW) {
                        T s = (T)0;
                        for (int j = 0; j < len; ++j) s = s + prod[lo + j];
                        ystage[r] = s;
                    } else {
                        is_long = 1;
                    }
                }
            }
            if (__syncthreads_or(is_long)) {
                for (int r = ra + wave; r < re; r += HIPK_THREADS / 64) {
                    const int lo = crowL[r] - g0;
                    const int hi2 = crowL[r + 1] - g0;
                    if (hi2 - lo > HIPK_LONG_ROW) {
                        T s = (T)0;
                        for (int j = lo + lane; j < hi2; j += 64) s = s + prod[j];
                        s = hipk_wave_sum(s);
                        if (lane == 0) ystage[r] = s;
                    }
                }
                __syncthreads();
            }
            ra = re;
        }
        if (t < nr) yrow = ystage[t];
    }

    // ---------------- output + fused dots ("tiled dot": one value per row, spec tree per tile)
    double d0 = 0.0, d1 = 0.0;
    if (t < nr) {
        T out = yrow;
        if (mode & HIPK_SPMV_RESID) out = brow - out;
        if (mode & HIPK_SPMV_SCALE) out = drow * out;
        y[r0 + t] = out;
        if (mode & HIPK_SPMV_DOT_W) d0 = (double)wrow * (double)out;
        if (mode & HIPK_SPMV_DOT_YY) d1 = (double)out * (double)out;
    }
    if (mode & (HIPK_SPMV_DOT_W | HIPK_SPMV_DOT_YY)) {
        // tile partial = ((s0 + s1) + (s2 + s3)), s_w = wavefront w's 64 values folded by v[l] += v[l+s], s = 32..1.
        // Each wavefront stores its s_w; the combine kernel forms the tile partial: no barrier here.
        if (mode & HIPK_SPMV_DOT_W) {
d0 = hipk_wave_sum(d0);
            if (lane == 0) a.tpart0[(size_t)tile * 4 + wave] = d0;
        }
        if (mode & HIPK_SPMV_DOT_YY) {
d1 = hipk_wave_sum(d1);
            if (lane == 0) a.tpart1[(size_t)tile * 4 + wave] = d1;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Long-row matrices (dense-as-CSR -- what the reference's own tests and benchmark feed, SURVEY fact 2 --,
// FEM blocks, power-law hubs): ROW PER WAVEFRONT.  Lanes stride over the row with coalesced col/val loads
// (4 steps in flight), accumulate in ascending order and fold with v[l] += v[l+s], s = 32..1: the long-row
// summation of the spec, so results are bit-identical to the tile kernel's.  Rows of <= HIPK_LONG_ROW entries
// are summed in CSR order by lane 0 (spec for short rows).
template <typename T>
__global__ __launch_bounds__(HIPK_THREADS) void hipk_spmv_rowwave_kernel(hipk_spmv_args a) {
    if (a.stop_it != nullptr && a.it >= *a.stop_it) return;
    const int lane = threadIdx.x & 63;
    const int64_t slot = (int64_t)blockIdx.x * (HIPK_THREADS / 64) + (threadIdx.x >> 6);
    const bool listed = a.row_list != nullptr;
    if (slot >= (listed ? (int64_t)a.n_list : a.n)) return;
    const int64_t row = listed ? (int64_t)a.row_list[slot] : slot;
    const int *__restrict__ col = a.col;
    const T *__restrict__ val = (const T *)a.val;
    const T *__restrict__ x = (const T *)a.x;
    const int lo = a.crow[row], hi = a.crow[row + 1];
    T s = (T)0;
    if (hi - lo <= HIPK_LONG_ROW) {
        if (lane == 0)
            for (int j = lo; j < hi; ++j) s = s + val[j] * x[col[j]];
    } else {
        int j = lo + lane;
        for (; j + 192 < hi; j += 256) {  // four strided steps in flight, added in ascending order
            const int c0 = col[j], c1 = col[j + 64], c2 = col[j + 128], c3 = col[j + 192];
            const T v0 = hipk_ld_nt(val + j), v1 = hipk_ld_nt(val + j + 64), v2 = hipk_ld_nt(val + j + 128),
                    v3 = hipk_ld_nt(val + j + 192);
            const T p0 = v0 * x[c0], p1 = v1 * x[c1], p2 = v2 * x[c2], p3 = v3 * x[c3];
            s = s + p0;
            s = s + p1;
            s = s + p2;
            s = s + p3;
        }
        for (; j < hi; j += 64) s = s + hipk_ld_nt(val + j) * x[col[j]];
        s = hipk_wave_sum(s);
    }
    if (lane == 0) {
        if (!listed && (a.mode & HIPK_SPMV_RESID)) s = ((const T *)a.bsub)[row] - s;  // pre-pass: raw sum, the tile kernel finishes
        if (!listed && (a.mode & HIPK_SPMV_SCALE)) s = ((const T *)a.dscale)[row] * s;
        ((T *)a.y)[row] = s;
    }
}

// fused-dot epilogue of the row-per-wavefront path: the tile kernel's per-wavefront sums, computed from y
template <typename T>
__global__ __launch_bounds__(HIPK_THREADS) void hipk_rowdot_kernel(hipk_spmv_args a) {
    if (a.stop_it != nullptr && a.it >= *a.stop_it) return;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int tile = blockIdx.x;
    const int64_t r = (int64_t)tile * HIPK_TILE + t;
    double d0 = 0.0, d1 = 0.0;
    if (r < a.n) {
        const T out = ((const T *)a.y)[r];
        if (a.mode & HIPK_SPMV_DOT_W) d0 = (double)((const T *)a.w)[r] * (double)out;
        if (a.mode & HIPK_SPMV_DOT_YY) d1 = (double)out * (double)out;
    }
    if (a.mode & HIPK_SPMV_DOT_W) {
d0 = hipk_wave_sum(d0);
        if (lane == 0) a.tpart0[(size_t)tile * 4 + wave] = d0;
    }
    if (a.mode & HIPK_SPMV_DOT_YY) {
d1 = hipk_wave_sum(d1);
        if (lane == 0) a.tpart1[(size_t)tile * 4 + wave] = d1;
    }
}

// chunk partial c = spec fold of the tile partials of chunk c (tiles_per_chunk = CH / 256).
// One WAVEFRONT per chunk (4 chunks per workgroup, no barriers): lane l plays the spec's virtual threads
// l, l+64, l+128, l+192 (each takes partials t, t+256, .. ascending), then the tree 128, 64, 32..1.
__device__ __forceinline__ double hipk_wave_fold(const double *__restrict__ tp, int cnt, int lane) {
    double a[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int t = lane + 64 * j;
        double acc = 0.0;
        for (int i = t; i < cnt; i += HIPK_THREADS) {
            const double *w4 = tp + (size_t)i * 4;  // the tile's four wavefront sums
            acc = acc + ((w4[0] + w4[1]) + (w4[2] + w4[3]));
        }
        a[j] = acc;
    }
    a[0] = a[0] + a[2];  // s = 128
    a[1] = a[1] + a[3];
    double v = a[0] + a[1];  // s = 64
v = hipk_wave_sum(v);
    return v;  // valid in lane 0
}

// Small systems (g <= 8 reduction chunks): the value of a fused dot, folded by the CONSUMER workgroup from the
// per-wavefront tile sums -- the combine kernel's chunk fold (one wavefront per chunk, two rounds for eight chunks) followed
// by the spec's fold of g <= 8 chunk partials (acc = 0.0 + part, then the tree's s = 4, 2, 1 steps): the bits a combine
// launch + hipk_reduce_parts would give.  All 256 threads must call it; cp: 8 doubles of LDS.
__device__ __forceinline__ double hipk_fold_tiles8(const double *__restrict__ tp, int ntiles, int tpc, int g, double *cp) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int c = wave; c < 8; c += HIPK_THREADS / 64) {
        double r = 0.0;
        if (c < g) {
            const int first = c * tpc;
            const int cnt = (ntiles - first < tpc) ? ntiles - first : tpc;
            r = hipk_wave_fold(tp + (size_t)first * 4, cnt, lane);
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

static __global__ __launch_bounds__(HIPK_THREADS) void hipk_tile_combine_kernel(const double *__restrict__ tp0,
                                                                                const double *__restrict__ tp1,
                                                                                double *__restrict__ part0,
                                                                                double *__restrict__ part1, int ntiles,
                                                                                int tiles_per_chunk, int nchunks,
                                                                                const int64_t *__restrict__ stop_it,
                                                                                int64_t it) {
    if (stop_it != nullptr && it >= *stop_it) return;
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * (HIPK_THREADS / 64) + (threadIdx.x >> 6);
    if (c >= nchunks) return;
    const int first = c * tiles_per_chunk;
    const int cntt = (ntiles - first < tiles_per_chunk) ? ntiles - first : tiles_per_chunk;
    if (tp0 != nullptr) {
        const double r = hipk_wave_fold(tp0 + (size_t)first * 4, cntt, lane);
        if (lane == 0) part0[c] = r;
    }
    if (tp1 != nullptr) {
        const double r = hipk_wave_fold(tp1 + (size_t)first * 4, cntt, lane);
        if (lane == 0) part1[c] = r;
    }
}
#endif

// host-side launcher (hipk_spmv.hip)
struct hipk_spmv_profiler;
int hipk_launch_spmv(const hipk_csr_s *h, const hipk_spmv_args &a, hipStream_t stream,
                     hipk_spmv_profiler *prof = nullptr);
