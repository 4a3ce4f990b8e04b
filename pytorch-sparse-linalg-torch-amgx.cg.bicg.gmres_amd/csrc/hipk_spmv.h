// hipk_spmv.h -- CSR SpMV for gfx950, LDS-staged ("stream") with fused dots.
//
// Replaces `torch.matmul(A, v)` (TSL:191) and, fused, the dot that follows it in
// every solver loop (TSL:846 <p,Ap>; TSL:910 <rhat,q>; TSL:926-930 <t,s>,<t,t>;
// TSL:352 ||A v||) and the residual form b - A x (TSL:820, 875, 791, 1008).
//
// Work decomposition (one workgroup of 256 threads per reduction chunk):
//   chunk  = CH rows (CH = 2048 * 2^k, <= 2048 chunks; hipk_common.h)
//   tile   = 256*VEC rows of the chunk (512 fp64 / 1024 fp32): row pointers staged in LDS
//   group  = a run of <= 256 rows of the tile whose nnz fit CAP LDS slots:
//            col/val are read fully coalesced (lane i <-> nnz j0+i), x is gathered
//            (L2 hits: XCD-aware chunk placement), the PRODUCTS go to LDS, then
//            thread r sums row r's products from LDS in CSR order (rows longer
//            than HIPK_LONG_ROW are summed by a whole wavefront, lanes strided)
//   output = thread t writes rows {VEC*t..VEC*t+VEC-1} of the tile as one 16-byte store
//            and accumulates the fused dots in exactly the order of the stand-alone
//            dot kernel (reduction spec), so fused and unfused dots are bit-identical.
// The kernel is HBM-bound: 12 B/nnz + 4 B/row of matrix, 8 B/row in, 8 B/row out (fp64).
#pragma once
#include "hipk_common.h"

#define HIPK_SPMV_DOT_W 1   // part0[c] = sum w_i * out_i
#define HIPK_SPMV_DOT_YY 2  // part1[c] = sum out_i * out_i
#define HIPK_SPMV_RESID 4   // out = bsub - A x  instead of A x
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
};

#ifdef __HIPCC__
// rows of the chunk whose row pointers are staged in LDS at once ("super-tile")
#define HIPK_SPMV_ST 2048

template <typename T, int CAP>
__global__ __launch_bounds__(HIPK_THREADS) void hipk_spmv_kernel(hipk_spmv_args a) {
    constexpr int VEC = hipk_vec<T>::VEC;
    constexpr int SUB = HIPK_THREADS * VEC;
    constexpr int NI = CAP / HIPK_THREADS;
    constexpr int ST = HIPK_SPMV_ST;
    static_assert(CAP % HIPK_THREADS == 0, "CAP must be a multiple of the workgroup size");
    static_assert(CAP * sizeof(T) >= 512 * sizeof(double), "prod[] doubles as reduction scratch");
    static_assert(ST % SUB == 0, "super-tile must hold whole output tiles");

    if (a.stop_it != nullptr && a.it >= *a.stop_it) return;
    const int c = hipk_xcd_chunk(blockIdx.x, a.g);
    if (c < 0) return;

    __shared__ __attribute__((aligned(16))) T prod[CAP];
    __shared__ __attribute__((aligned(16))) T ytile[SUB];
    __shared__ int crowL[ST + 1];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    const int *__restrict__ crow = a.crow;
    const int *__restrict__ col = a.col;
    const T *__restrict__ val = (const T *)a.val;
    const T *__restrict__ x = (const T *)a.x;
    T *__restrict__ y = (T *)a.y;
    const T *__restrict__ w = (const T *)a.w;
    const T *__restrict__ bsub = (const T *)a.bsub;
    const int mode = a.mode;

    const int64_t row0 = (int64_t)c * a.ch;
    const int64_t row1 = (row0 + a.ch < a.n) ? row0 + a.ch : a.n;
    double acc0 = 0.0, acc1 = 0.0;

    // group end: largest re in (ra, lim] with at most CAP entries; re == ra <=> row ra alone exceeds CAP
    auto find_group = [&](int ra, int lim) -> int {
        const int j0 = crowL[ra];
        int hi = (ra + HIPK_THREADS < lim) ? ra + HIPK_THREADS : lim;
        if (crowL[hi] - j0 <= CAP) return hi;
        int lo = ra;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (crowL[mid] - j0 <= CAP) lo = mid; else hi = mid;
        }
        return lo;
    };

    for (int64_t st0 = row0; st0 < row1; st0 += ST) {
        const int nst = (int)((row1 - st0 < ST) ? (row1 - st0) : ST);
        __syncthreads();  // previous super-tile fully consumed before crowL is overwritten
        for (int i = t; i <= nst; i += HIPK_THREADS) crowL[i] = crow[st0 + i];
        __syncthreads();

        // ---- software pipeline over groups: col/val of group g+1 are in flight while
        // group g gathers x, multiplies, and is reduced from LDS.
        int tile0 = 0;                                  // first row of the current output tile
        int tile1 = (SUB < nst) ? SUB : nst;            // end of the current output tile
        int ra = 0;
        int re = find_group(ra, tile1);
        int j0 = crowL[ra];
        int cnt = (re > ra) ? crowL[re] - j0 : 0;
        int cc[NI];
        T vv[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int j = t + i * HIPK_THREADS;
            if (j < cnt) {
                cc[i] = col[j0 + j];
                vv[i] = val[j0 + j];
            }
        }
        while (true) {
            // -- bounds of the next group (may open the next output tile)
            const int ra2 = (re > ra) ? re : ra + 1;
            const bool tile_done = (ra2 >= tile1);
            const bool has_next = (ra2 < nst);
            int tile1n = tile1;
            if (tile_done) tile1n = (tile1 + SUB < nst) ? tile1 + SUB : nst;
            int re2 = ra2, j02 = 0, cnt2 = 0;
            if (has_next) {
                re2 = find_group(ra2, tile1n);
                j02 = crowL[ra2];
                cnt2 = (re2 > ra2) ? crowL[re2] - j02 : 0;
            }
            // -- gather x for the current group, then put the next group's loads in flight
            T xv[NI];
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int j = t + i * HIPK_THREADS;
                if (j < cnt) xv[i] = x[cc[i]];
            }
            int cc2[NI];
            T vv2[NI];
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int j = t + i * HIPK_THREADS;
                if (j < cnt2) {
                    cc2[i] = col[j02 + j];
                    vv2[i] = val[j02 + j];
                }
            }
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int j = t + i * HIPK_THREADS;
                if (j < cnt) prod[j] = vv[i] * xv[i];
            }
            if (re == ra) {
                // one row with more than CAP entries: a single wavefront streams it from global
                // memory, lanes strided by 64, same summation tree as a long LDS row
                if (wave == 0) {
                    const int j1 = crowL[ra + 1];
                    T s = (T)0;
                    for (int j = j0 + lane; j < j1; j += 64) s = s + val[j] * x[col[j]];
#pragma unroll
                    for (int o = 32; o >= 1; o >>= 1) s = s + __shfl_down(s, o);
                    if (lane == 0) ytile[ra - tile0] = s;
                }
            }
            __syncthreads();

            int is_long = 0;
            {
                const int r = ra + t;
                if (r < re) {
                    const int lo = crowL[r] - j0;
                    const int len = crowL[r + 1] - j0 - lo;
                    if (len <= HIPK_LONG_ROW) {
                        T s = (T)0;
                        for (int j = 0; j < len; ++j) s = s + prod[lo + j];
                        ytile[r - tile0] = s;
                    } else {
                        is_long = 1;
                    }
                }
            }
            if (__syncthreads_or(is_long)) {
                for (int r = ra + wave; r < re; r += HIPK_THREADS / 64) {
                    const int lo = crowL[r] - j0;
                    const int hi2 = crowL[r + 1] - j0;
                    if (hi2 - lo > HIPK_LONG_ROW) {
                        T s = (T)0;
                        for (int j = lo + lane; j < hi2; j += 64) s = s + prod[j];
#pragma unroll
                        for (int o = 32; o >= 1; o >>= 1) s = s + __shfl_down(s, o);
                        if (lane == 0) ytile[r - tile0] = s;
                    }
                }
                __syncthreads();
            }

            if (tile_done) {
                // ---- output + fused dots, in reduction-spec order
                const int ns = tile1 - tile0;
                const int i0 = VEC * t;
                if (i0 < ns) {
                    const int64_t gi = st0 + tile0 + i0;
                    T out[VEC];
                    const int nv = (ns - i0 < VEC) ? ns - i0 : VEC;
#pragma unroll
                    for (int k = 0; k < VEC; ++k) out[k] = (k < nv) ? ytile[i0 + k] : (T)0;
                    if (mode & HIPK_SPMV_RESID) {
#pragma unroll
                        for (int k = 0; k < VEC; ++k)
                            if (k < nv) out[k] = bsub[gi + k] - out[k];
                    }
                    if (nv == VEC) {
                        typename hipk_vec<T>::type ov;
                        T *op = (T *)&ov;
#pragma unroll
                        for (int k = 0; k < VEC; ++k) op[k] = out[k];
                        *(typename hipk_vec<T>::type *)(y + gi) = ov;
                    } else {
                        for (int k = 0; k < nv; ++k) y[gi + k] = out[k];
                    }
                    if (mode & HIPK_SPMV_DOT_W) {
#pragma unroll
                        for (int k = 0; k < VEC; ++k)
                            if (k < nv) acc0 = fma((double)w[gi + k], (double)out[k], acc0);
                    }
                    if (mode & HIPK_SPMV_DOT_YY) {
#pragma unroll
                        for (int k = 0; k < VEC; ++k)
                            if (k < nv) acc1 = fma((double)out[k], (double)out[k], acc1);
                    }
                }
                tile0 = tile1;
                tile1 = tile1n;
                // the next group's ytile writes come after its own barrier, i.e. after every
                // thread has finished reading ytile here.
            }
            if (!has_next) break;
            ra = ra2;
            re = re2;
            j0 = j02;
            cnt = cnt2;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                cc[i] = cc2[i];
                vv[i] = vv2[i];
            }
        }
    }

    if (mode & (HIPK_SPMV_DOT_W | HIPK_SPMV_DOT_YY)) {
        __syncthreads();
        hipk_block_sum2(acc0, acc1, (double *)prod);
        if (t == 0) {
            if (mode & HIPK_SPMV_DOT_W) a.part0[c] = acc0;
            if (mode & HIPK_SPMV_DOT_YY) a.part1[c] = acc1;
        }
    }
}
#endif

// host-side launcher (hipk_spmv.hip)
int hipk_launch_spmv(const hipk_csr_s *h, const hipk_spmv_args &a, hipStream_t stream);
