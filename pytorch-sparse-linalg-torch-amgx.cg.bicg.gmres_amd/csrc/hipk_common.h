// hipk_common.h -- shared host/device definitions of libhipk (gfx950 only).
//
// Reduction spec (mirrored bit-for-bit by oracle/krylov_oracle.c):
//   * a vector of n elements is cut in G = ceil(n/CH) chunks, CH = 2048 * 2^k with
//     the smallest k such that G <= 2048;
//   * inside a chunk, "virtual thread" t (0..255) owns elements
//     {VEC*t .. VEC*t+VEC-1} + 256*VEC*j, j = 0,1,..  (VEC = 16 B / sizeof(T)) and
//     accumulates acc = fma(a_i, b_i, acc) over them in ascending i, in fp64;
//   * the 256 accumulators are summed by the tree  v[t] += v[t+s], s = 128,64,..,1;
//   * the G chunk partials are summed by thread t taking partials t, t+256, ..
//     in ascending order, then the same tree.
// Element-wise updates use mul-then-add (two roundings) exactly like the
// reference's `_add(x, _mul(alpha, p))`; the library is built with
// -ffp-contract=off so nothing is fused unless `fma` is spelled out.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "hipk.h"

#define HIPK_THREADS 256
#define HIPK_MAX_PARTS 2048
#define HIPK_BASE_CHUNK 2048
// number of partial-sum slots (of HIPK_MAX_PARTS doubles each) in a scratch buffer
#define HIPK_SCRATCH_SLOTS 4

// ---------------------------------------------------------------- error plumbing
void hipk_set_error(const char *fmt, ...);

#define HIPK_CHECK_HIP(expr)                                                          \
    do {                                                                              \
        hipError_t _e = (expr);                                                       \
        if (_e != hipSuccess) {                                                       \
            hipk_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr,              \
                           hipGetErrorString(_e));                                    \
            return HIPK_ERR_HIP;                                                      \
        }                                                                             \
    } while (0)

#define HIPK_REQUIRE(cond, code, msg)                                                 \
    do {                                                                              \
        if (!(cond)) {                                                                \
            hipk_set_error("%s:%d: %s", __FILE__, __LINE__, msg);                     \
            return (code);                                                            \
        }                                                                             \
    } while (0)

static inline bool hipk_aligned16(const void *p) { return (((uintptr_t)p) & 15u) == 0; }

// ---------------------------------------------------------------- chunk geometry
struct hipk_geom {
    int64_t n;
    int ch;  // chunk size (elements)
    int g;   // chunk count (<= HIPK_MAX_PARTS)
};

static inline hipk_geom hipk_make_geom(int64_t n) {
    hipk_geom gm;
    gm.n = n;
    const int64_t full = (int64_t)HIPK_BASE_CHUNK * HIPK_MAX_PARTS;
    int64_t q = (n + full - 1) / full;
    if (q < 1) q = 1;
    int64_t p = 1;
    while (p < q) p <<= 1;
    gm.ch = (int)(HIPK_BASE_CHUNK * p);
    gm.g = (int)((n + gm.ch - 1) / gm.ch);
    if (gm.g < 1) gm.g = 1;
    return gm;
}

// ---------------------------------------------------------------- CSR handle
// what hipk_launch_spmv resolved for one (mode, chunk size, chunk count) of a sliced-ELL handle: kernel instantiation, grid, walk.
// Resolved once (environment switches, two occupancy queries, the kernel's name) and reused by every later launch (ADVICE r2:
// that host work sat on the per-iteration path of the launch-bound sizes); hipk_csr_set_path drops the plans,
// HIPK_SPMV_NO_PLAN_CACHE=1 resolves per launch again (in-process A/B probes that flip the switches between launches)
struct hipk_spmv_plan {
    int mode, ch, g;
    void *kern;
    int lgrid, group_tiles;
    bool chunked, strided;
    char name[96];
};

struct hipk_csr_s {
    int64_t n_rows, n_cols, nnz;
    int dtype;        // hipk_dtype
    int *crow;        // device, n_rows+1, owned
    int *col;         // device, nnz, owned
    const void *val;  // device, nnz, borrowed
    hipk_geom geom;   // chunking of the row space
    int device;
    // pinned host word block used by solves to follow the device loop: words 0-1 are the poller's read slots,
    // word 2 is the signal word the loop's deciding kernel stores to (hipk_pacer, hipk_solve.h)
    int64_t *host_poll;  // hipHostMalloc, 16 x int64
    double *tile_part;   // device, 2 x 4 x ceil(n_rows/256): per-wavefront sums of the fused dots (4 per tile)
    int *huge_rows;      // device, owned or null: rows with more entries than the LDS product buffer, handled by a
    int n_huge;          //   row-per-wavefront pre-pass when the matrix as a whole is short-rowed
    int max_row_len;     // structure analysis at creation
    int max_tile_nnz;    //   (tile = 256 consecutive rows)
    // window plan of the one-launch CG loop (hipk_mid.h: hipk_mid_plan_get): per row block, the 256-column tiles its rows reference
    void *mid_plan_mem;  // device, owned: tlo | nslot | tiles | map | needed | status
    int mid_plan_state;  // 0 not computed, 1 usable, -1 the matrix does not fit (range or tile count)
    int mid_plan_nch, mid_plan_nblk, mid_plan_max_slots;
    // coded form (hipk_coded.h), present when the matrix has <= 256 distinct (col - row, value) pairs and short rows
    unsigned char *code;    // device, nnz (+32 bytes of padding), owned
    unsigned char *rowlen;  // device, n_rows, owned
    int *dict_off;          // device, 256
    void *dict_val;         // device, 256 values of `dtype`
    int n_codes;            // 0: no coded form
    int path_override;      // hipk_csr_set_path: 0 auto, 1 never use the coded form
    int coded_layout;       // 1: codes in CSR order (+ rowlen); 2: sliced-ELL planes (tile_off / sell_w);
                            // 3: sliced-ELL planes of OFFSET codes + value planes (sell_vals)
    void *sell_vals;        // device, sell_bytes values of `dtype`, owned (layout 3)
    int *tile_off;          // device, ntiles + 1 (sliced-ELL)
    unsigned long long *tile_ucode;  // device, ntiles: code bytes shared by all 256 rows of a tile (0: rows differ);
    int n_uniform_tiles;             //   null unless enough tiles are uniform (constant-coefficient stencils)
    int uniform_units;               //   256-byte units of code planes the uniform tiles own (never read by the SpMV)
    // round 3: tiles whose rows differ only by WHICH entries of one pattern they have (grid-line ends of a stencil): the
    // two-rows-per-lane kernel takes them like uniform tiles, a per-row presence mask skipping the absent entries
    unsigned long long *tile_wcode;  // device, ntiles: tile_ucode for uniform tiles; union pattern with byte 7 = HIPK_SELL_MASKED for
                                     //   masked tiles; 0 otherwise (null: no such analysis)
    unsigned char *row_mask;         // device, 256 * ntiles: bit k = entry k of the tile's pattern is present in this row
    int n_masked_tiles, masked_units;
    int sell_w;             // uniform tile size in units of 256 B, or 0
    int64_t sell_bytes;     // bytes of all tiles
    int sell_loop;          // persistent sliced-ELL kernel: grid = sell_loop * 8 * n_cu workgroups (0: off)
    int n_cu;               // compute units of the device
    int sell_chunked;       // 1: the persistent kernel may take one reduction chunk per workgroup (no combine launch)
    // matrix-free operator (hipk_op_create): every product is op_cb(op_user, x, y) + an epilogue kernel; no CSR arrays
    int (*op_cb)(void *user, const void *x_dev, void *y_dev);
    void *op_user;
    mutable hipk_spmv_plan plans[12];
    mutable int n_plans;
};

#ifdef __HIPCC__
// MI355X admits 256-thread workgroups per CU up to min(API answer, 8, floor(800 / (ceil(sgpr / 16) * 16 + 16))): at 82-96 scalar
// registers only SEVEN, although the occupancy API and the compiler's "Occupancy" remark say 8 (MI355X_MICROARCH.md, "Residency").
// Round 3 found the GMRES multi-dot (96 SGPRs), the fused-exchange CG kernels (83-84) and the 8-wide two-rows-per-lane SpMV
// (83-106) in that band -- a grid of 1954 chunks then runs as 1792 + a second round of 162.  Kernels that are budgeted for eight
// workgroups per CU carry this attribute; tests/test_kernel_resources.py reads the compiler's report for VGPRs AND SGPRs.
#define HIPK_SGPR80 __attribute__((amdgpu_num_sgpr(80)))
// ---------------------------------------------------------------- device helpers
template <typename T>
struct hipk_vec;
template <>
struct hipk_vec<double> {
    typedef double2 type;
    static constexpr int VEC = 2;
};
template <>
struct hipk_vec<float> {
    typedef float4 type;
    static constexpr int VEC = 4;
};

// XCD-aware placement: workgroups b and b+8 share an XCD (round-robin dispatch), so
// give XCD k the k-th contiguous eighth of the chunks: the x-vector lines a 5-point
// stencil re-reads (+-1 grid line) then hit in that XCD's own L2.
// Speed only -- correctness never depends on it.  Returns -1 for padding blocks.
__device__ __forceinline__ int hipk_xcd_chunk(int b, int g) {
    const int per = (g + 7) >> 3;
    const int c = (b & 7) * per + (b >> 3);
    return (c < g && (b >> 3) < per) ? c : -1;
}
static inline int hipk_xcd_grid(int g) { return ((g + 7) >> 3) << 3; }

// v of lane i + N of the same 16-lane row (DPP row_shl): the strides 8, 4, 2, 1 of the wavefront sum's tree without the
// LDS crossbar of ds_bpermute.  Same pairing as __shfl_down for the lanes the tree reads, so the same bits.
template <int N>
__device__ __forceinline__ double hipk_row_shl(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x100 + N, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x100 + N, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
// the spec's wavefront sum (strides 32 ... 1), result valid in lane 0
// v of lane i + 32 (lanes 0..31) / lane i + 16 (lanes 0..15): gfx950's v_permlane32_swap / v_permlane16_swap with both
// operands the same register -- the second result holds the upper half / the odd 16-lane rows moved down
// (tools/ubench/permlane_probe.hip prints the mapping).
__device__ __forceinline__ double hipk_lane_up32(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    return __hiloint2double(__builtin_amdgcn_permlane32_swap(hi, hi, false, false)[1],
                            __builtin_amdgcn_permlane32_swap(lo, lo, false, false)[1]);
}
__device__ __forceinline__ double hipk_lane_up16(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    return __hiloint2double(__builtin_amdgcn_permlane16_swap(hi, hi, false, false)[1],
                            __builtin_amdgcn_permlane16_swap(lo, lo, false, false)[1]);
}
__device__ __forceinline__ double hipk_wave_sum(double d) {
    d = d + hipk_lane_up32(d);
    d = d + hipk_lane_up16(d);
    d = d + hipk_row_shl<8>(d);
    d = d + hipk_row_shl<4>(d);
    d = d + hipk_row_shl<2>(d);
    d = d + hipk_row_shl<1>(d);
    return d;
}

// the same tree for an fp32 value (row sums of the row-per-wavefront SpMV with fp32 storage)
template <int N>
__device__ __forceinline__ float hipk_row_shl(float v) {
    const int i = __float_as_int(v);
    return __int_as_float(__builtin_amdgcn_update_dpp(i, i, 0x100 + N, 0xF, 0xF, false));
}
__device__ __forceinline__ float hipk_wave_sum(float d) {
    int i = __float_as_int(d);
    d = d + __int_as_float(__builtin_amdgcn_permlane32_swap(i, i, false, false)[1]);
    i = __float_as_int(d);
    d = d + __int_as_float(__builtin_amdgcn_permlane16_swap(i, i, false, false)[1]);
    d = d + hipk_row_shl<8>(d);
    d = d + hipk_row_shl<4>(d);
    d = d + hipk_row_shl<2>(d);
    d = d + hipk_row_shl<1>(d);
    return d;
}

// Sum of the 256 per-thread values with the spec's tree. Result valid in ALL threads.
// sbuf: 256 doubles of LDS. Leaves sbuf reusable (trailing barrier).
__device__ __forceinline__ double hipk_block_sum(double v, double *sbuf) {
    const int t = threadIdx.x;
    sbuf[t] = v;
    __syncthreads();
    if (t < 128) sbuf[t] = sbuf[t] + sbuf[t + 128];
    __syncthreads();
    if (t < 64) {
        const double a = hipk_wave_sum(sbuf[t] + sbuf[t + 64]);  // strides 32 ... 1, register moves only
        if (t == 0) sbuf[0] = a;
    }
    __syncthreads();
    const double r = sbuf[0];
    __syncthreads();
    return r;
}

// Two sums at once (sbuf: 512 doubles).
__device__ __forceinline__ void hipk_block_sum2(double &v0, double &v1, double *sbuf) {
    const int t = threadIdx.x;
    sbuf[t] = v0;
    sbuf[256 + t] = v1;
    __syncthreads();
    if (t < 128) {
        sbuf[t] = sbuf[t] + sbuf[t + 128];
        sbuf[256 + t] = sbuf[256 + t] + sbuf[256 + t + 128];
    }
    __syncthreads();
    if (t < 64) {
        double a = sbuf[t] + sbuf[t + 64];
        double b = sbuf[256 + t] + sbuf[256 + t + 64];
        a = hipk_wave_sum(a);
        b = hipk_wave_sum(b);
        if (t == 0) {
            sbuf[0] = a;
            sbuf[256] = b;
        }
    }
    __syncthreads();
    v0 = sbuf[0];
    v1 = sbuf[256];
    __syncthreads();
}

// Final reduction of g (<= 2048) chunk partials, redundantly in every workgroup:
// all workgroups obtain the same bits without any inter-workgroup hand-off inside a
// launch (the partials were written by the PREVIOUS kernel on the stream).
__device__ __forceinline__ double hipk_reduce_parts(const double *__restrict__ part, int g,
                                                    double *sbuf) {
    const int t = threadIdx.x;
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < HIPK_MAX_PARTS / HIPK_THREADS; ++k) {
        const int i = t + k * HIPK_THREADS;
        if (i < g) acc = acc + part[i];
    }
    return hipk_block_sum(acc, sbuf);
}

__device__ __forceinline__ void hipk_reduce_parts2(const double *__restrict__ p0,
                                                   const double *__restrict__ p1, int g,
                                                   double &r0, double &r1, double *sbuf) {
    const int t = threadIdx.x;
    double a = 0.0, b = 0.0;
#pragma unroll
    for (int k = 0; k < HIPK_MAX_PARTS / HIPK_THREADS; ++k) {
        const int i = t + k * HIPK_THREADS;
        if (i < g) {
            a = a + p0[i];
            b = b + p1[i];
        }
    }
    hipk_block_sum2(a, b, sbuf);
    r0 = a;
    r1 = b;
}
#endif  // __HIPCC__
