/*
 * hipk.h -- C ABI of libhipk.so: MI355X (gfx950) kernels for the Module-A
 * iterative-solver hot path (CSR SpMV, deterministic dots, fused CG / BiCGStab /
 * GMRES updates, device-resident solve loops).
 *
 * The reference (Litianyu141/Pytorch-Sparse-Linalg-torch-amgx.cg.bicg.gmres) has
 * NO native boundary: its hot path is Python calling ATen.  Every entry point
 * below therefore names the reference Python construct it replaces
 * (TSL = src/pytorch_sparse_solver/module_a/torch_sparse_linalg.py).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch types.
 *   - every pointer named *_dev / x / y / b / work is a DEVICE pointer
 *     (tensor.data_ptr()); vectors must be 16-byte aligned.
 *   - every function returns HIPK_OK (0) or a negative hipk_status;
 *     hipk_last_error() gives the thread-local message of the last failure.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).
 *   - the library never allocates inside a solve: the caller supplies `work`
 *     (size from hipk_*_work_bytes), so solves are graph/stream friendly.
 *   - a handle owns scratch its kernels write (per-tile sums of the fused dots):
 *     calls on ONE handle must be ordered on one stream (or by events);
 *     different handles are independent.
 *   - all reductions are atomic-free with a fixed summation tree
 *     (DESIGN.md "reduction spec"): results are bitwise run-to-run reproducible
 *     and bitwise equal to oracle/krylov_oracle.c.
 */
#ifndef HIPK_H
#define HIPK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HIPK_VERSION 300

typedef struct hipk_csr_s *hipk_csr_t;
typedef void *hipk_stream_t; /* hipStream_t */

enum hipk_dtype { HIPK_F32 = 0, HIPK_F64 = 1 };

enum hipk_status {
    HIPK_OK = 0,
    HIPK_ERR_ARG = -1,         /* bad argument (null pointer, negative size, ...) */
    HIPK_ERR_HIP = -2,         /* a HIP runtime call failed                       */
    HIPK_ERR_ALIGN = -3,       /* vector pointer not 16-byte aligned              */
    HIPK_ERR_UNSUPPORTED = -4, /* dtype / option not built                        */
    HIPK_ERR_WORKSPACE = -5,   /* work buffer too small                           */
    HIPK_ERR_NO_DEVICE = -6    /* no gfx950 device visible                        */
};

/* GMRES least-squares variants, TSL:755-760 (`solve_method`). */
enum hipk_gmres_method { HIPK_GMRES_BATCHED = 0, HIPK_GMRES_INCREMENTAL = 1 };

/* Solver parameters: the keyword arguments of cg/bicgstab/gmres
 * (TSL:1019-1021, 1091-1093, 641-644). */
typedef struct {
    double tol;            /* relative tolerance (python float, rounded through fp32 as the reference does) */
    double atol;           /* absolute tolerance                                                           */
    int64_t maxiter;       /* CG/BiCGStab: iterations; GMRES: restart cycles. <0 => 10*n (TSL:982-984)      */
    int32_t restart;       /* GMRES Krylov dimension (TSL:642)                                             */
    int32_t gmres_method;  /* hipk_gmres_method                                                            */
    int32_t check_every;   /* stream-ordered polling interval of the fallback pacing (<=0: default), see below */
    int32_t gpu_tolerances;/* 1: GMRES uses the `device.type=='cuda'` tolerance branch (TSL:737-740)       */
    int32_t profile;       /* 1: time every SpMV launch of the loop (start/stop events bound to the dispatch) and report
                              stats.spmv_ms_avg; CG only: 2 = the update kernel, 3 = the direction kernel (large systems:
                              its flat-grid launch), 4 = the scalars launch that precedes the flat-grid one           */
    int32_t reserved;
} hipk_params;

/* Side channel the reference does not have: filled on return. */
typedef struct {
    int64_t iterations;     /* CG/BiCGStab loop iterations; GMRES restart cycles                    */
    int64_t matvecs;        /* SpMV launches that did work                                          */
    int32_t info;           /* 0 converged / -1 not, decided exactly as TSL:1007-1016 / TSL:766-773 */
    int32_t breakdown;      /* BiCGStab: 0, -10 (rho), -11 (alpha/omega) as TSL:902-936; GMRES: 1 on happy breakdown */
    double b_norm;          /* ||b||                                                                */
    double residual_norm;   /* true ||b - A x|| recomputed after the loop                           */
    double x_norm;          /* ||x|| (NaN check)                                                    */
    double threshold;       /* the value residual_norm was compared against                         */
    double recurrence_rs;   /* last recurrence <r,r> (CG: gamma)                                    */
    double solve_ms;        /* device time of the whole solve, HIP events on `stream`               */
    double spmv_ms_avg;     /* average of stop - start over the launches params.profile selects (start/stop events bound to the
                               dispatch, hipExtLaunchKernel); RAW: nothing subtracted.  The start stamp is taken when the dispatch
                               is picked up, 0.6-1.5 us before its first wave when the previous kernel is still draining
                               (csrc/hipk_solve.h has the comparison with rocprofv3's averages). */
    int64_t spmv_profiled;  /* number of SpMV launches in that average                              */
    double dispatch_span_ms_avg; /* the same figure (in the diagnostic chain mode spmv_ms_avg is stop-to-stop instead; version 200:
                                    event_overhead_ms) */
} hipk_stats;

int hipk_version(void);
/* 16 hex digits: sha1 over the sources this library was compiled from (csrc/Makefile).  Counter profiles under profiles/ carry it;
 * bench.py quotes a `traffic` figure only when it was taken on the build that is running. */
const char *hipk_build_id(void);
const char *hipk_last_error(void);

/* Number of visible HIP devices whose arch is gfx950 (0 => nothing can run). */
int hipk_device_count(void);

/* ---- CSR handle --------------------------------------------------------------
 * Replaces the tensor `A` captured by `_normalize_matvec` (TSL:176-208).
 * crow/col are device arrays of idx_bytes (4 or 8) wide integers as torch stores
 * them (int64); they are narrowed once to int32 (SURVEY 7.2 "index width").
 * `val` is BORROWED: it must stay alive and unchanged until hipk_csr_destroy.
 * Column indices must be sorted within each row (torch CSR invariant) only for
 * bitwise parity with the oracle, not for correctness. */
int hipk_csr_create(hipk_csr_t *out, int64_t n_rows, int64_t n_cols, int64_t nnz,
                    const void *crow_dev, const void *col_dev, int idx_bytes,
                    const void *val_dev, int dtype, hipk_stream_t stream);
int hipk_csr_destroy(hipk_csr_t h);
int64_t hipk_csr_rows(hipk_csr_t h);
int64_t hipk_csr_nnz(hipk_csr_t h);
/* Algorithmic bytes of one SpMV: nnz*(sizeof(val)+4)+(n+1)*4+2*n*sizeof(val). */
int64_t hipk_csr_spmv_bytes(hipk_csr_t h);

/* Which SpMV kernel family the structure analysis of hipk_csr_create selected.  All of them produce the same
 * bits (one summation spec, oracle/krylov_oracle.c).
 *   TILE_FAST : 256-row tiles, every tile fits the LDS product buffer and no row exceeds 32 entries
 *   TILE      : 256-row tiles with the general path (long rows, dense tiles, huge-row pre-pass)
 *   ROWWAVE   : row per wavefront (mean row length >= 48: dense-as-CSR, FEM blocks)
 *   CODED     : the matrix has at most 256 distinct (col - row, value) pairs and short rows (finite-difference /
 *               finite-volume stencils, e.g. matrix_utils.py:193-257 and ldc_solver_common.py:90-135): the handle
 *               keeps one byte per entry + one byte per row + the dictionary and streams those instead of
 *               col/val/crow.  The values are SNAPSHOTTED at creation (`val` must not change anyway, see above).
 *   OFFSET_CODED : too many distinct values, but at most 255 distinct column OFFSETS and short rows (variable-coefficient
 *               stencils): one offset-code byte + the value per entry in coalesced planes, 9 instead of 12 bytes
 *               per entry and no row pointers.  Values snapshotted as for CODED. */
enum hipk_spmv_path { HIPK_PATH_TILE_FAST = 0, HIPK_PATH_TILE = 1, HIPK_PATH_ROWWAVE = 2, HIPK_PATH_CODED = 3,
                      HIPK_PATH_OFFSET_CODED = 4 };
int hipk_csr_spmv_path(hipk_csr_t h);
/* Name of the kernel instantiation the calling thread's most recent SpMV launch selected (stand-alone or inside a solver),
 * as the profiler prints it, e.g. "hipk_spmv_sell_wide_kernel<5,1>"; "" before the first launch.  Measurement scripts label
 * their per-kernel figures with it instead of guessing the dispatch. */
const char *hipk_last_spmv_kernel(void);
/* mode 0: automatic (default); 1: never use the coded forms (A/B measurements, parity tests).
 * Environment: HIPK_SPMV_CODED=0 at creation time skips building the coded forms altogether,
 * HIPK_SPMV_OFFSET_CODED=0 only the offset-coded one. */
int hipk_csr_set_path(hipk_csr_t h, int mode);
/* Bytes one SpMV has to move in the format the selected path streams (= hipk_csr_spmv_bytes unless CODED). */
int64_t hipk_csr_format_bytes(hipk_csr_t h);

/* ---- transpose (adjoint solves) ------------------------------------------------
 * The implicit-diff backward of the reference solves with A^T (`ImplicitAdjointFunction.backward`, TSL:1237-1248; `A.T`,
 * TSL:1245).  hipk_csr_transpose writes the CSR arrays of A^T -- int32 row pointers [n_cols + 1], int32 columns [nnz]
 * sorted within each row, values [nnz] of the handle's dtype -- into caller-owned device buffers, from which a handle
 * for A^T is created with hipk_csr_create (idx_bytes 4).  Device-side stable sort by column; `work` >=
 * hipk_csr_transpose_work_bytes(h), 256-byte aligned.  Setup step, not on the per-iteration path. */
size_t hipk_csr_transpose_work_bytes(hipk_csr_t h);
int hipk_csr_transpose(hipk_csr_t h, int32_t *crow_t_dev, int32_t *col_t_dev, void *val_t_dev, void *work,
                       size_t work_bytes, hipk_stream_t stream);

/* ---- reduction geometry ------------------------------------------------------
 * Every dot/norm is a two-level fixed tree: the vector is cut in `count` chunks
 * of `chunk` elements (chunk = 2048 * 2^k, count <= 2048); see DESIGN.md. */
int hipk_chunk_size(int64_t n);
int hipk_chunk_count(int64_t n);

/* ---- primitives --------------------------------------------------------------
 * scratch_dev: device buffer of hipk_scratch_bytes() bytes (partial sums).      */
size_t hipk_scratch_bytes(void);

/* y = A x  (torch.matmul(A, v), TSL:191). */
int hipk_spmv(hipk_csr_t h, const void *x, void *y, hipk_stream_t stream);
/* y = A x and out_dev[0] = <w, y>  (TSL:845-846 fused). */
int hipk_spmv_dot(hipk_csr_t h, const void *x, void *y, const void *w,
                  double *out_dev, void *scratch_dev, hipk_stream_t stream);
/* out_dev[0] = <x, y>  (`_vdot_real_tree`, TSL:130-139). Accumulates in fp64. */
int hipk_dot(int64_t n, const void *x, const void *y, int dtype, double *out_dev,
             void *scratch_dev, hipk_stream_t stream);
/* y = a*x + y with mul-then-add rounding (`_add(y, _mul(a, x))`, TSL:847). */
int hipk_axpy(int64_t n, double a, const void *x, void *y, int dtype, hipk_stream_t stream);
/* y = x + b*y  (`_add(z, _mul(beta, p))`, TSL:852). */
int hipk_xpby(int64_t n, const void *x, double b, void *y, int dtype, hipk_stream_t stream);

/* ---- whole solves (device-resident loops) ------------------------------------
 * x: in = x0, out = solution.  b is not modified.  `work` >= *_work_bytes.
 * The loop stops at exactly the iteration the reference stops at (device-side
 * stop word: launches past it are no-ops).  The host follows the loop through a
 * pinned word the deciding kernel stores to and keeps a few iterations queued
 * ahead; HIPK_HOST_SIGNAL=0 (or a word that stops moving) selects stream-ordered
 * reads of the stop word every check_every iterations instead.
 * A handle owns scratch that its solves and fused-dot products share (per-tile
 * partial sums, the pinned signal words): run ONE whole solve / hipk_spmv_ex
 * with dot modes per handle at a time; plain hipk_spmv calls may overlap.      */
size_t hipk_cg_work_bytes(int64_t n, int dtype);
/* `_isolve(_cg_solve)`: TSL:806-856 + 968-1016. */
int hipk_cg_solve(hipk_csr_t A, const void *b, void *x, void *work, size_t work_bytes,
                  const hipk_params *prm, hipk_stats *st, hipk_stream_t stream);

size_t hipk_bicgstab_work_bytes(int64_t n, int dtype);
/* `_isolve(_bicgstab_solve)`: TSL:859-964 + 968-1016. */
int hipk_bicgstab_solve(hipk_csr_t A, const void *b, void *x, void *work, size_t work_bytes,
                        const hipk_params *prm, hipk_stats *st, hipk_stream_t stream);

size_t hipk_gmres_work_bytes(int64_t n, int restart, int dtype);
/* `gmres`: TSL:641-803 with `_gmres_batched` (TSL:431-493) or
 * `_gmres_incremental` (TSL:557-638). */
int hipk_gmres_solve(hipk_csr_t A, const void *b, void *x, void *work, size_t work_bytes,
                     const hipk_params *prm, hipk_stats *st, hipk_stream_t stream);

/* ---- CG with a Jacobi preconditioner ---------------------------------
 * `cg(A, b, M=...)` of the reference (TSL:1019-1021) with M(v) = dinv .* v, the iteration of TSL:806-856 for a
 * non-identity M: gamma = <r, M r>, stop test on <r,r>, `info` from ||M (b - A x)|| (TSL:1007).  `dinv` is a
 * device vector of the matrix' dtype (the reciprocal diagonal); the scaling is fused into the update and direction
 * kernels.  Same params / stats / work-buffer conventions as hipk_cg_solve. */
size_t hipk_pcg_work_bytes(int64_t n, int dtype);
int hipk_pcg_solve(hipk_csr_t A, const void *dinv, const void *b, void *x, void *work, size_t work_bytes,
                   const hipk_params *prm, hipk_stats *st, hipk_stream_t stream);
/* GMRES with the same M (left preconditioning: the reference applies M after every A, TSL:351, 791, 766; ptol from
 * ||M b||, TSL:750).  The row scaling runs in the SpMV epilogue.  Work buffer: hipk_gmres_work_bytes. */
int hipk_pgmres_solve(hipk_csr_t A, const void *dinv, const void *b, void *x, void *work, size_t work_bytes,
                      const hipk_params *prm, hipk_stats *st, hipk_stream_t stream);
/* BiCGStab with the same M, applied BEFORE A as the reference does (phat = M p, shat = M s, TSL:908, 922; x advances
 * with phat / shat, TSL:942; `info` from ||M (b - A x)||): phat and shat are two more work vectors. */
size_t hipk_pbicgstab_work_bytes(int64_t n, int dtype);
int hipk_pbicgstab_solve(hipk_csr_t A, const void *dinv, const void *b, void *x, void *work, size_t work_bytes,
                         const hipk_params *prm, hipk_stats *st, hipk_stream_t stream);
/* Placement probe for systems whose vectors live in HBM (N >> 8 M rows): the memory shape of the CG direction step (reads r, p, x;
 * writes p, x) on the three vectors, storing back the bits it loaded (safe on live data); *us_out = the fastest of `reps` (<= 16)
 * timed passes in microseconds.  40 n bytes (fp64) per pass.  On MI355X such a step runs at one of two discrete speeds depending on
 * where the allocation landed physically; the Python host re-draws the work allocation when it reads the slow one (_hipk.py). */
int hipk_placement_probe(int64_t n, const void *r, void *p, void *x, int dtype, int reps, double *us_out, hipk_stream_t stream);

/* ---- MATRIX-FREE operators (the reference's `_normalize_matvec` takes a callable for all three solvers, TSL:176-208).
 * hipk_op_create makes a handle WITHOUT a matrix: every product y = A x of a solve is `op(user, x_dev, y_dev)`, which enqueues
 * y = A(x) (vectors of `dtype`, n elements) on the solve's stream and returns 0.  The residual form b - A x, the row scaling of
 * the Jacobi forms and the fused dots of the SpMV kernels follow in one epilogue kernel with the SAME reduction spec ("tiled dot"),
 * so hipk_cg_solve / hipk_bicgstab_solve / hipk_gmres_solve (and the hipk_p* forms) run unchanged on such a handle -- device
 * stop word, one host synchronisation per GMRES cycle and none inside CG / BiCGStab -- and, when `op` computes this library's
 * SpMV of a matrix, return that matrix's solve bit for bit.  The host calls `op` when it ENQUEUES an iteration (a few ahead of the
 * device): it must not synchronise, and it may be called for a few iterations past the stop (their results are never read).
 * The one-launch small-system kernels need the matrix and are not taken.  Destroy with hipk_csr_destroy. */
typedef int (*hipk_op_fn)(void *user, const void *x_dev, void *y_dev);
int hipk_op_create(hipk_csr_t *out, int64_t n, int dtype, hipk_op_fn op, void *user, hipk_stream_t stream);

/* BiCGStab with the CALLER's preconditioner: M(user, in_dev, out_dev) enqueues out = M(in) (vectors of the handle's
 * dtype, n elements) on `stream` and returns 0; it is called for p and s of every iteration (TSL:908, 922) and once
 * for the final residual (TSL:1007).  Workspace as hipk_pbicgstab_work_bytes.  With M = diag(dinv) the iterates equal
 * hipk_pbicgstab_solve's bit for bit. */
typedef int (*hipk_precond_fn)(void *user, const void *in_dev, void *out_dev);
int hipk_pbicgstab_solve_cb(hipk_csr_t A, hipk_precond_fn M, void *user, const void *b, void *x, void *work,
                            size_t work_bytes, const hipk_params *prm, hipk_stats *st, hipk_stream_t stream);
/* GMRES with the CALLER's preconditioner, applied after every A (left preconditioning: v = M(A v) TSL:351,
 * M(b - A x) TSL:791/766, ptol from ||M b|| TSL:750).  M is called in place (in_dev == out_dev) on workspace
 * vectors.  Workspace as hipk_gmres_work_bytes. */
int hipk_pgmres_solve_cb(hipk_csr_t A, hipk_precond_fn M, void *user, const void *b, void *x, void *work,
                         size_t work_bytes, const hipk_params *prm, hipk_stats *st, hipk_stream_t stream);

/* ---- block-Jacobi preconditioner ------------------------------------------
 * out = M in with M = blockdiag(A)^-1: `binv_dev` holds the inverted block_size x block_size diagonal blocks, row-major
 * per block, ceil(n / block_size) of them (a ragged last block is padded with identity rows/columns).  The device kernel
 * behind `BlockJacobiPreconditioner`, a callable for the reference's `M` hook (TSL:849, 908, 922, 351) that cg / bicgstab /
 * gmres run between their fused kernels.  z_i is an fma chain over the block's columns in ascending order. */
int hipk_block_jacobi_apply(int64_t n, int block_size, const void *binv_dev, const void *in, void *out, int dtype,
                            hipk_stream_t stream);

/* ---- step API: externally driven loops (row-partitioned multi-GPU CG) ------------
 * The reference is single-device; the row-partitioned solver (north_star) drives the
 * SAME fused kernels from the host side of each rank and exchanges (a) the x-vector
 * halo and (b) the chunk partial sums between launches.  Rows are partitioned on
 * reduction-chunk boundaries of the GLOBAL problem, so every rank reduces the
 * all-gathered partials in the same fixed order: results are bitwise identical to the
 * single-GPU solve for any rank count.
 *   chunk_rows : chunk size of the GLOBAL row count (hipk_chunk_size(n_global))
 *   g_red      : number of partials of ALL ranks (hipk_chunk_count(n_global))
 * Kernels index partial OUTPUTS by local chunk and read partial INPUTS from the
 * gathered (global) arrays. */
int hipk_csr_create_ex(hipk_csr_t *out, int64_t n_rows, int64_t n_cols, int64_t nnz,
                       const void *crow_dev, const void *col_dev, int idx_bytes,
                       const void *val_dev, int dtype, int chunk_rows, hipk_stream_t stream);
/* mode bits: 1 part0[c] = <w, out>; 2 part1[c] = <out, out>; 4 out = bsub - A x */
int hipk_spmv_ex(hipk_csr_t h, const void *x, void *y, int mode, const void *w, const void *bsub,
                 double *part0, double *part1, const int64_t *stop_dev, int64_t it,
                 hipk_stream_t stream);
int hipk_dot_parts(int64_t n, int chunk_rows, const void *x, const void *y, int dtype,
                   double *part, hipk_stream_t stream);
/* out_dev[0] = fixed-order sum of part[0..g)  (the second level of every dot) */
int hipk_reduce_parts(const double *part_dev, int g, double *out_dev, hipk_stream_t stream);
/* dst[i] = src[idx[i]], i < m  (halo pack) */
int hipk_gather(int64_t m, const int32_t *idx_dev, const void *src, void *dst, int dtype,
                hipk_stream_t stream);
size_t hipk_cg_scal_bytes(void); /* device scalar block: {gamma[2], atol2, bs, res2, xx, stop_it(int64), host signal ptr (null here)} */
int hipk_cg_start(int64_t n_local, int chunk_rows, int g_red, void *scal_dev, const double *part_rr,
                  const double *part_bb, const void *r, void *p, int dtype, double tol, double atol,
                  int64_t maxiter, hipk_stream_t stream);
/* r -= alpha Ap, partials of <r,r>   (alpha = gamma / sum(part_pAp)) */
int hipk_cg_update(int64_t n_local, int chunk_rows, int g_red, const void *scal_dev, int64_t it,
                   const double *part_pAp, const void *Ap, void *r, double *part_rr_out, int dtype,
                   hipk_stream_t stream);
/* x += alpha p alone (alpha = gamma / sum(part_pAp), the bits hipk_cg_update / hipk_cg_direction derive): for a caller that
 * runs it on a side stream while a collective is in flight and then calls hipk_cg_direction with x = NULL */
int hipk_cg_xupdate(int64_t n_local, int chunk_rows, int g_red, const void *scal_dev, int64_t it, const double *part_pAp,
                    const void *p, void *x, int dtype, hipk_stream_t stream);
/* x += alpha p; p = r + beta p; gamma <- <r,r>; stop test   (the x update rides on the pass over p; x = NULL: p only) */
int hipk_cg_direction(int64_t n_local, int chunk_rows, int g_red, void *scal_dev, int64_t it,
                      int64_t maxiter, const double *part_pAp, const double *part_rr, const void *r, void *p,
                      void *x, int dtype, hipk_stream_t stream);

/* CG with a CALLABLE preconditioner (`M` of cg(), TSL:821, 849): the host runs
 *   hipk_spmv_ex(p -> Ap, <p,Ap>) | hipk_cg_update | z = M(r) (the caller's own device code, same stream) |
 *   hipk_dot_parts(r, z) | hipk_cgm_direction
 * gamma = <r,z> steers alpha / beta, the stop test uses <r,r> (TSL:835-841). */
int hipk_cgm_start(int64_t n_local, int chunk_rows, int g_red, void *scal_dev, const double *part_rz,
                   const double *part_rr, const double *part_bb, const void *z, void *p, int dtype,
                   double tol, double atol, int64_t maxiter, hipk_stream_t stream);
/* x += alpha p; p = z + beta p; gamma <- <r,z>; stop test on <r,r> */
int hipk_cgm_direction(int64_t n_local, int chunk_rows, int g_red, void *scal_dev, int64_t it, int64_t maxiter,
                       const double *part_pAp, const double *part_rz, const double *part_rr, const void *z,
                       void *p, void *x, int dtype, hipk_stream_t stream);

/* ---- row-partitioned CG, the whole loop of one rank driven from C (north_star: "the SpMV shards row-blocks across the
 * GPUs with an RCCL allgather of the x-vector halo and an allreduce for the global dot") ----------------------------------
 * The collectives are called through function pointers the caller resolves from the librccl that created `comm`
 * (signatures of ncclGroupStart / ncclGroupEnd / ncclAllGather / ncclSend / ncclRecv; datatype 8 = ncclFloat64), on the
 * solver's stream.  Tests plug in host-staged stand-ins. */
typedef struct {
    int (*group_start)(void);
    int (*group_end)(void);
    int (*all_gather)(const void *send, void *recv, size_t count, int datatype, void *comm, void *stream);
    int (*send)(const void *buf, size_t count, int datatype, int peer, void *comm, void *stream);
    int (*recv)(void *buf, size_t count, int datatype, int peer, void *comm, void *stream);
    void *comm;
    void *fused;   /* NULL, or a hipk_p2p_t created with a fused area (hipk_p2p_create2): hipk_dist_cg_solve then folds the two
                      exchanges of an iteration into its update / direction kernels (csrc/hipk_fx.h) -- no collective launch inside
                      the loop; the entry points above still serve the set-up and the final residual */
} hipk_rccl;

/* One rank's view of the partition (pytorch_sparse_solver/distributed.py: RowPartition + HaloPlan). */
typedef struct {
    int32_t rank, world;
    int64_t n_local;   /* rows this rank owns (> 0 on every rank)                                        */
    int64_t n_ext;     /* n_local + number of halo entries: length of x, p, r                            */
    int64_t n_global;
    int32_t chunk_rows; /* reduction chunk of the GLOBAL problem (hipk_chunk_size(n_global))              */
    int32_t g_red;      /* partials of all ranks (hipk_chunk_count(n_global))                             */
    int32_t per;        /* chunks per rank, the all-gather count (per * world >= g_red)                   */
    int32_t halo_mode;  /* 1: neighbour send/recv pairs; 0: all-gather of slabs padded to `slab` entries  */
    int32_t n_send;     /* owned entries other ranks need, grouped by destination rank                    */
    int32_t n_ghost;    /* = n_ext - n_local                                                              */
    int32_t slab;       /* halo_mode 0: padded slab length                                                */
    int32_t reserved;
    const int32_t *send_idx_dev;  /* [n_send] local row of each packed entry                              */
    const int32_t *ghost_src_dev; /* [n_ghost] halo_mode 0: position of each halo entry in the gathered slabs */
    const int32_t *send_off_dev;  /* device, [world + 1] or NULL: bounds of each destination's group in send_idx_dev (fused exchanges) */
    const int64_t *dest_off_dev;  /* device, [world] or NULL: where this rank's group starts in each destination's ghost tail     */
    const int32_t *send_counts;   /* host, [world]: entries sent to each rank                             */
    const int32_t *recv_counts;   /* host, [world]: halo entries received from each rank (in rank order)  */
    const int64_t *send_first;    /* host, [world] or NULL: halo_mode 1, >= 0 where the entries for that rank are the
                                     CONTIGUOUS local rows send_first[p] .. + send_counts[p] (row blocks of a stencil):
                                     they are sent straight from the vector, and when that holds for every peer the pack
                                     kernel is not launched at all                                       */
} hipk_dist_plan;

size_t hipk_dist_cg_work_bytes(const hipk_dist_plan *plan);
/* `A_local`: this rank's row block with columns renumbered to [0, n_ext) (hipk_csr_create_ex with the global chunk size).
 * x_ext: n_ext doubles, x0 in the first n_local on entry, the solution there on return.  params.check_every = batch
 * size of the loop (default 16).  Same `info` rule and, bit for bit, the same iterates as hipk_cg_solve on the whole
 * system (TSL:806-856, 968-1016). */
int hipk_dist_cg_solve(hipk_csr_t A_local, const hipk_dist_plan *plan, const hipk_rccl *coll, const void *b_local,
                       void *x_ext, void *work, size_t work_bytes, const hipk_params *prm, hipk_stats *st,
                       hipk_stream_t stream);

/* Row-partitioned BiCGStab (TSL:859-964 via `_isolve`): the same conventions and the same plan / collective structs; x_ext carries
 * x0 / the solution in its first n_local entries.  Bit for bit the iterates, counts and breakdown codes of hipk_bicgstab_solve on
 * the whole system.  Five collective launches per iteration (the all-gathers of the six dots' partials, the halos of p and s). */
size_t hipk_dist_bicgstab_work_bytes(const hipk_dist_plan *plan);
int hipk_dist_bicgstab_solve(hipk_csr_t A_local, const hipk_dist_plan *plan, const hipk_rccl *coll, const void *b_local,
                             void *x_ext, void *work, size_t work_bytes, const hipk_params *prm, hipk_stats *st,
                             hipk_stream_t stream);

/* Row-partitioned GMRES (TSL:641-803; params.restart <= 31, params.gmres_method, params.gpu_tolerances as hipk_gmres_solve): the
 * kernels of the large-system path with in-place all-gathers of their chunk partials and the halo of v_k before each SpMV; bit for
 * bit the iterates, cycle and operator-application counts of hipk_gmres_solve on the whole system.  per * world <= 2048. */
size_t hipk_dist_gmres_work_bytes(const hipk_dist_plan *plan, int restart);
int hipk_dist_gmres_solve(hipk_csr_t A_local, const hipk_dist_plan *plan, const hipk_rccl *coll, const void *b_local,
                          void *x_ext, void *work, size_t work_bytes, const hipk_params *prm, hipk_stats *st,
                          hipk_stream_t stream);

/* ---- EXPERIMENTAL peer-to-peer exchange provider for the loop above (csrc/hipk_p2p.hip) ---------------------------------
 * Each rank owns a device mailbox that every peer maps through HIP IPC; an all-gather is ONE small kernel per rank (publish
 * blocks store into the peers' mailboxes, collect blocks wait on per-source sequence flags).  No reference counterpart.
 * Protocol: create on every rank -> export the 64-byte handle -> exchange the handles out of band (rank order) -> connect.
 * hipk_p2p_group_start/_group_end/_all_gather have the signatures of the hipk_rccl members (comm = the hipk_p2p_t);
 * leave hipk_rccl.send/.recv NULL and hipk_dist_plan.halo_mode 0.  max_count = the largest `count` of any call. */
typedef struct hipk_p2p_s *hipk_p2p_t;
int hipk_p2p_create(hipk_p2p_t *out, int rank, int world, size_t max_count);
/* ... with a FUSED AREA for a partition of fx_per chunks per rank and at most fx_ghost_cap ghost entries on any rank (the same two
 * numbers on every rank): per-source sequence flags, the gathered partial arrays and the ghost tail of both exchanges of a CG
 * iteration, written by the peers' update / direction kernels themselves (csrc/hipk_fx.h; hipk_rccl.fused). */
int hipk_p2p_create2(hipk_p2p_t *out, int rank, int world, size_t max_count, int fx_per, int fx_ghost_cap);
int hipk_p2p_export(hipk_p2p_t c, void *handle64);
int hipk_p2p_connect(hipk_p2p_t c, const void *handles /* world x 64 bytes */);
int hipk_p2p_destroy(hipk_p2p_t c);
int hipk_p2p_error(hipk_p2p_t c); /* 1: a wait gave up (a peer never published) since the last query -- the results of the calls in
                                     between are void; the query clears the flag */
int hipk_p2p_group_start(void);
int hipk_p2p_group_end(void);
int hipk_p2p_all_gather(const void *send, void *recv, size_t count, int datatype, void *comm, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* HIPK_H */
