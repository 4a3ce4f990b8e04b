/*
 * krylov_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C; fp64, and with -DORC_F32 the fp32-storage extension: vectors and
 * matrix values in float, every dot accumulated in fp64, scalars in fp64) of the reference's
 * Module-A hot path:
 *   cg / bicgstab / gmres of
 *   /root/reference/src/pytorch_sparse_solver/module_a/torch_sparse_linalg.py  (= TSL)
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product (libhipk.so + the Python package) never does.
 *
 * Parity pinning: the reference ships no golden vectors (its tests are residual
 * thresholds only), so this restatement is pinned against outputs of the reference
 * itself, generated in the build container by oracle/gen_golden.py and committed
 * under tests/golden/ (inputs + x, info, matvec counts, residuals).
 * tests/test_oracle_golden.py checks every fixture.
 *
 * The arithmetic ORDER is part of the contract with the HIP kernels (DESIGN.md
 * "reduction spec"): they reproduce this file bit-for-bit.
 *   - dot: chunks of CH = 2048*2^k elements (<= 2048 chunks); in a chunk "virtual
 *     thread" t of 256 owns elements {2t,2t+1} + 512 j and accumulates with fma in
 *     ascending order; 256 accumulators are folded by v[t] += v[t+s], s = 128..1;
 *     chunk partials are folded the same way (thread t takes partials t, t+256, ..).
 *   - "tiled dot" (every dot FUSED into an SpMV epilogue: <p,Ap>, <rhat,q>, <t,s>, <t,t>,
 *     ||A v||^2, ||b - A x||^2): rows are cut in tiles of 256; thread t of a tile forms the
 *     rounded product a_i*b_i of row t (0 beyond n); each wavefront's 64 products are folded
 *     by v[l] += v[l+s], s = 32..1, and the TILE partial is ((s0+s1)+(s2+s3)); the tile partials of a chunk are folded like chunk
 *     partials (thread t takes t, t+256, .. then the tree) into the chunk partial.
 *   - SpMV row: products rounded, then added in CSR order (rows <= 32 entries);
 *     longer rows: 64 strided lane sums folded by v[l] += v[l+s], s = 32..1.
 *   - element-wise updates: multiply, round, add, round (as `_add(x, _mul(a, p))`).
 * Build with -ffp-contract=off (oracle/Makefile) so nothing is fused implicitly.
 *
 * Threading (orc_set_threads) only parallelises over chunks / rows, which are
 * independent in the spec, so results are bitwise identical for any thread count.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_THREADS 256
#define ORC_MAX_PARTS 2048
#define ORC_BASE_CHUNK 2048
#define ORC_LONG_ROW 32
#define ORC_MAXM 127 /* largest GMRES restart (the reference accepts any, TSL:641-644) */
#define ORC_LD 128
#define ORC_EPS64 2.220446049250313e-16 /* torch.finfo(torch.float64).eps */
#define ORC_EPS32 1.1920928955078125e-07 /* torch.finfo(torch.float32).eps */
#ifdef ORC_F32
typedef float real;
#define ORC_EPS ORC_EPS32 /* guards use the eps of the working dtype; the GMRES absolute floor keeps ORC_EPS64 */
#else
typedef double real;
#define ORC_EPS ORC_EPS64
#endif
#define ORC_VEC (16 / (int)sizeof(real)) /* elements a virtual thread owns per step: 16-byte accesses */
#define ORC_INV_SQRT2 0.7071067811865476 /* TSL:63 */

#ifdef ORC_F32 /* second compilation of this file with -DORC_F32: fp32 storage, symbols orc32_* */
#define orc_set_threads orc32_set_threads
#define orc_get_threads orc32_get_threads
#define orc_chunk_geom orc32_chunk_geom
#define orc_dot_parts orc32_dot_parts
#define orc_dot_parts_ch orc32_dot_parts_ch
#define orc_reduce_parts orc32_reduce_parts
#define orc_dot orc32_dot
#define orc_dot_tiled_parts_ch orc32_dot_tiled_parts_ch
#define orc_dot_tiled orc32_dot_tiled
#define orc_spmv orc32_spmv
#define orc_cg orc32_cg
#define orc_pcg_jacobi orc32_pcg_jacobi
#define orc_block_jacobi_apply orc32_block_jacobi_apply
#define orc_pcg_blockjacobi orc32_pcg_blockjacobi
#define orc_bicgstab orc32_bicgstab
#define orc_bicgstab_jacobi orc32_bicgstab_jacobi
#define orc_gmres orc32_gmres
#define orc_gmres_jacobi orc32_gmres_jacobi
#endif

static int g_threads = 1;

void orc_set_threads(int t) {
    g_threads = t < 1 ? 1 : t;
#ifdef _OPENMP
    omp_set_num_threads(g_threads);
#endif
}
int orc_get_threads(void) { return g_threads; }

typedef struct {
    int64_t iterations; /* cg/bicgstab iterations, gmres restart cycles */
    int64_t matvecs;
    int32_t info;
    int32_t breakdown;
    double b_norm;
    double residual_norm;
    double x_norm;
    double threshold;
    double recurrence_rs;
} orc_stats;

/* ------------------------------------------------------------------ geometry */
void orc_chunk_geom(int64_t n, int *ch, int *g) {
    const int64_t full = (int64_t)ORC_BASE_CHUNK * ORC_MAX_PARTS;
    int64_t q = (n + full - 1) / full;
    if (q < 1) q = 1;
    int64_t p = 1;
    while (p < q) p <<= 1;
    *ch = (int)(ORC_BASE_CHUNK * p);
    *g = (int)((n + *ch - 1) / *ch);
    if (*g < 1) *g = 1;
}

static double tree256(double *v) {
    for (int s = 128; s >= 1; s >>= 1)
        for (int t = 0; t < s; ++t) v[t] = v[t] + v[t + s];
    return v[0];
}

static double reduce_parts(const double *part, int g) {
    double v[ORC_THREADS];
    for (int t = 0; t < ORC_THREADS; ++t) {
        double acc = 0.0;
        for (int k = 0; k < ORC_MAX_PARTS / ORC_THREADS; ++k) {
            const int i = t + k * ORC_THREADS;
            if (i < g) acc = acc + part[i];
        }
        v[t] = acc;
    }
    return tree256(v);
}

static double chunk_dot(const real *a, const real *b, int64_t base, int64_t end) {
    double v[ORC_THREADS];
    for (int t = 0; t < ORC_THREADS; ++t) v[t] = 0.0;
    /* element e of the chunk belongs to virtual thread (e mod 256 VEC)/VEC, VEC = 16 B / sizeof(real)
       (2 for fp64, 4 for fp32); ascending e is ascending order inside every thread */
    for (int64_t i = base; i < end; ++i) {
        const int t = (int)(((i - base) % (ORC_THREADS * ORC_VEC)) / ORC_VEC);
        v[t] = fma((double)a[i], (double)b[i], v[t]);
    }
    return tree256(v);
}

/* G chunk partials of <a,b> (also what a rank of the row-partitioned solver owns) */
void orc_dot_parts(int64_t n, const real *a, const real *b, double *parts) {
    int ch, g;
    orc_chunk_geom(n, &ch, &g);
#pragma omp parallel for schedule(static) if (g_threads > 1)
    for (int c = 0; c < g; ++c) {
        const int64_t base = (int64_t)c * ch;
        const int64_t end = base + ch < n ? base + ch : n;
        parts[c] = (base < end) ? chunk_dot(a, b, base, end) : 0.0;
    }
}

/* same with an explicit chunk size: the partials a rank of the row-partitioned solver owns */
void orc_dot_parts_ch(int64_t n, int ch, const real *a, const real *b, double *parts) {
    const int g = (int)((n + ch - 1) / ch);
#pragma omp parallel for schedule(static) if (g_threads > 1)
    for (int c = 0; c < g; ++c) {
        const int64_t base = (int64_t)c * ch;
        const int64_t end = base + ch < n ? base + ch : n;
        parts[c] = (base < end) ? chunk_dot(a, b, base, end) : 0.0;
    }
}

double orc_reduce_parts(const double *parts, int g) { return reduce_parts(parts, g); }

/* `_vdot_real_tree` (TSL:130-139) */
double orc_dot(int64_t n, const real *a, const real *b) {
    int ch, g;
    orc_chunk_geom(n, &ch, &g);
    double *parts = (double *)malloc(sizeof(double) * (size_t)g);
    orc_dot_parts(n, a, b, parts);
    const double r = reduce_parts(parts, g);
    free(parts);
    return r;
}

/* chunk partials of the tiled dot (what hipk_spmv_kernel + hipk_tile_combine_kernel produce) */
void orc_dot_tiled_parts_ch(int64_t n, int ch, const real *a, const real *b, double *parts) {
    const int g = (int)((n + ch - 1) / ch);
    const int tpc = ch / 256;
#pragma omp parallel for schedule(static) if (g_threads > 1)
    for (int c = 0; c < g; ++c) {
        double tp[ORC_MAX_PARTS];
        const int64_t base = (int64_t)c * ch;
        const int64_t end = base + ch < n ? base + ch : n;
        int nt = 0;
        for (int64_t t0 = base; t0 < end; t0 += 256, ++nt) {
            double v[ORC_THREADS], sw[4];
            for (int t = 0; t < 256; ++t) v[t] = (t0 + t < end) ? (double)a[t0 + t] * (double)b[t0 + t] : 0.0;
            for (int w = 0; w < 4; ++w) { /* one wavefront: v[l] += v[l+s], s = 32..1 */
                double *u = v + 64 * w;
                for (int s = 32; s >= 1; s >>= 1)
                    for (int l = 0; l < s; ++l) u[l] = u[l] + u[l + s];
                sw[w] = u[0];
            }
            tp[nt] = (sw[0] + sw[1]) + (sw[2] + sw[3]);
        }
        (void)tpc;
        parts[c] = reduce_parts(tp, nt);
    }
}

double orc_dot_tiled(int64_t n, const real *a, const real *b) {
    int ch, g;
    orc_chunk_geom(n, &ch, &g);
    double *parts = (double *)malloc(sizeof(double) * (size_t)g);
    orc_dot_tiled_parts_ch(n, ch, a, b, parts);
    const double r = reduce_parts(parts, g);
    free(parts);
    return r;
}

/* ------------------------------------------------------------------ SpMV */
static real row_sum(const int32_t *col, const real *val, const real *x, int lo, int hi) {
    const int len = hi - lo;
    if (len <= ORC_LONG_ROW) {
        real s = (real)0;
        for (int j = lo; j < hi; ++j) {
            const real p = val[j] * x[col[j]];
            s = s + p;
        }
        return s;
    }
    real v[64];
    for (int l = 0; l < 64; ++l) {
        real s = (real)0;
        for (int j = lo + l; j < hi; j += 64) {
            const real p = val[j] * x[col[j]];
            s = s + p;
        }
        v[l] = s;
    }
    for (int s = 32; s >= 1; s >>= 1)
        for (int l = 0; l < s; ++l) v[l] = v[l] + v[l + s];
    return v[0];
}

/* y = A x  (`torch.matmul(A, v)`, TSL:191); bsub != NULL: y = bsub - A x (TSL:820) */
void orc_spmv(int64_t n, const int32_t *crow, const int32_t *col, const real *val, const real *x,
              const real *bsub, real *y) {
#pragma omp parallel for schedule(static) if (g_threads > 1)
    for (int64_t r = 0; r < n; ++r) {
        const real s = row_sum(col, val, x, crow[r], crow[r + 1]);
        y[r] = bsub ? bsub[r] - s : s;
    }
}

/* ------------------------------------------------------------------ helpers */
static double tmax(double a, double b) { /* torch.maximum: NaN wins */
    if (isnan(a) || isnan(b)) return NAN;
    return a > b ? a : b;
}
static double tmin(double a, double b) {
    if (isnan(a) || isnan(b)) return NAN;
    return a < b ? a : b;
}
static double norm_from_sq(double v) { return sqrt(v < 0.0 ? 0.0 : v); } /* `_norm`, TSL:154-162 */

typedef struct {
    int64_t n;
    const int32_t *crow, *col;
    const real *val;
} csr_t;

static void isolve_epilogue(const csr_t *A, const real *b, const real *x, double tol, double atol,
                            double bs, real *tmp, orc_stats *st) {
    /* TSL:1007-1016 */
    orc_spmv(A->n, A->crow, A->col, A->val, x, b, tmp);
    st->residual_norm = norm_from_sq(orc_dot_tiled(A->n, tmp, tmp)); /* fused in the SpMV */
    st->b_norm = norm_from_sq(bs);
    st->x_norm = norm_from_sq(orc_dot(A->n, x, x));
    st->threshold = tmax((double)(float)tol * st->b_norm, (double)(float)atol);
    st->info = (isnan(st->x_norm) || st->residual_norm > st->threshold) ? -1 : 0;
}

/* ------------------------------------------------------------------ CG: TSL:806-856 via _isolve TSL:968-1016 */
int orc_cg(int64_t n, const int32_t *crow, const int32_t *col, const real *val, const real *b,
           real *x /* in: x0, out: x */, double tol, double atol, int64_t maxiter, orc_stats *st) {
    csr_t A = {n, crow, col, val};
    memset(st, 0, sizeof(*st));
    if (maxiter < 0) maxiter = 10 * n;
    real *r = (real *)malloc(sizeof(real) * (size_t)n);
    real *p = (real *)malloc(sizeof(real) * (size_t)n);
    real *Ap = (real *)malloc(sizeof(real) * (size_t)n);
    const double bs = orc_dot(n, b, b);
    const float tolf = (float)tol, atolf = (float)atol; /* torch.tensor(python float) is fp32 */
    const double a2 = (double)(tolf * tolf) * bs, a3 = (double)(atolf * atolf);
    const double atol2 = a2 > a3 ? a2 : a3;
    orc_spmv(n, crow, col, val, x, b, r);
    int64_t matvecs = 1;
    double gamma = orc_dot_tiled(n, r, r); /* fused in the residual SpMV */
    memcpy(p, r, sizeof(real) * (size_t)n);
    int64_t k = 0;
    while (!(k >= maxiter || gamma <= atol2)) {
        orc_spmv(n, crow, col, val, p, NULL, Ap);
        ++matvecs;
        const double pAp = orc_dot_tiled(n, p, Ap); /* fused in the SpMV */
        const double alpha = gamma / pAp;
#pragma omp parallel for schedule(static) if (g_threads > 1)
        for (int64_t i = 0; i < n; ++i) {
            const real m0 = (real)alpha * p[i];
            x[i] = x[i] + m0;
            const real m1 = (real)alpha * Ap[i];
            r[i] = r[i] - m1;
        }
        const double rr = orc_dot(n, r, r);
        const double beta = rr / gamma;
#pragma omp parallel for schedule(static) if (g_threads > 1)
        for (int64_t i = 0; i < n; ++i) {
            const real m = (real)beta * p[i];
            p[i] = r[i] + m;
        }
        gamma = rr;
        ++k;
    }
    isolve_epilogue(&A, b, x, tol, atol, bs, Ap, st);
    st->iterations = k;
    st->matvecs = matvecs + 1;
    st->recurrence_rs = gamma;
    free(r);
    free(p);
    free(Ap);
    return 0;
}

/* ------------------------------------------------------------------ CG with M = diag(dinv) (Jacobi), TSL:806-856
 * with `M is not _identity`: z = M r, gamma = <r,z>, the stop test uses rs = <r,r> (TSL:835-838), and the final
 * `info` compares ||M (b - A x)|| (TSL:1007).  Restates the device kernels of hipk_pcg_solve:
 *   start      z = dinv*r (one rounding), p = z, gamma0 = plain chunked dot <r,z>; rs0 = tiled dot of the residual SpMV
 *   update     r -= alpha*Ap (two roundings), z = dinv*r, plain chunked dots <r,r> and <r,z>
 *   direction  x += alpha*p, z = dinv*r again (same bits), p = z + beta*p                                            */
static double chunk_dot_scaled(const real *a, const real *d, int with_a, int64_t base, int64_t end) {
    /* with_a = 1: sum fma(a_i, (real)(d_i*a_i), .)   (<r, z>);   with_a = 0: sum fma(m_i, m_i, .), m_i = (real)(d_i*a_i) */
    double v[ORC_THREADS];
    for (int t = 0; t < ORC_THREADS; ++t) v[t] = 0.0;
    for (int64_t i = base; i < end; ++i) {
        const int t = (int)(((i - base) % (ORC_THREADS * ORC_VEC)) / ORC_VEC);
        const real m = d[i] * a[i];
        v[t] = with_a ? fma((double)a[i], (double)m, v[t]) : fma((double)m, (double)m, v[t]);
    }
    return tree256(v);
}

static double dot_scaled(int64_t n, const real *a, const real *d, int with_a) {
    int ch, g;
    orc_chunk_geom(n, &ch, &g);
    double parts[ORC_MAX_PARTS];
#pragma omp parallel for schedule(static) if (g_threads > 1)
    for (int c = 0; c < g; ++c) {
        const int64_t base = (int64_t)c * ch;
        const int64_t end = base + ch < n ? base + ch : n;
        parts[c] = chunk_dot_scaled(a, d, with_a, base, end);
    }
    return reduce_parts(parts, g);
}

int orc_pcg_jacobi(int64_t n, const int32_t *crow, const int32_t *col, const real *val, const real *dinv,
                   const real *b, real *x /* in: x0, out: x */, double tol, double atol, int64_t maxiter,
                   orc_stats *st) {
    memset(st, 0, sizeof(*st));
    if (maxiter < 0) maxiter = 10 * n;
    real *r = (real *)malloc(sizeof(real) * (size_t)n);
    real *p = (real *)malloc(sizeof(real) * (size_t)n);
    real *Ap = (real *)malloc(sizeof(real) * (size_t)n);
    const double bs = orc_dot(n, b, b);
    const float tolf = (float)tol, atolf = (float)atol;
    const double a2 = (double)(tolf * tolf) * bs, a3 = (double)(atolf * atolf);
    const double atol2 = a2 > a3 ? a2 : a3;
    orc_spmv(n, crow, col, val, x, b, r);
    int64_t matvecs = 1;
    double rs = orc_dot_tiled(n, r, r); /* fused in the residual SpMV */
    for (int64_t i = 0; i < n; ++i) p[i] = dinv[i] * r[i];
    double gamma = dot_scaled(n, r, dinv, 1);
    int64_t k = 0;
    while (!(k >= maxiter || rs <= atol2)) {
        orc_spmv(n, crow, col, val, p, NULL, Ap);
        ++matvecs;
        const double pAp = orc_dot_tiled(n, p, Ap);
        const double alpha = gamma / pAp;
#pragma omp parallel for schedule(static) if (g_threads > 1)
        for (int64_t i = 0; i < n; ++i) {
            const real m1 = (real)alpha * Ap[i];
            r[i] = r[i] - m1;
        }
        const double rr = orc_dot(n, r, r);
        const double rz = dot_scaled(n, r, dinv, 1);
        const double beta = rz / gamma;
#pragma omp parallel for schedule(static) if (g_threads > 1)
        for (int64_t i = 0; i < n; ++i) {
            const real m0 = (real)alpha * p[i];
            x[i] = x[i] + m0;
            const real z = dinv[i] * r[i];
            const real m = (real)beta * p[i];
            p[i] = z + m;
        }
        gamma = rz;
        rs = rr;
        ++k;
    }
    /* TSL:1007-1016 with M: ||M (b - A x)|| */
    orc_spmv(n, crow, col, val, x, b, Ap);
    ++matvecs;
    st->residual_norm = norm_from_sq(dot_scaled(n, Ap, dinv, 0));
    st->b_norm = norm_from_sq(bs);
    st->x_norm = norm_from_sq(orc_dot(n, x, x));
    st->threshold = tmax((double)(float)tol * st->b_norm, (double)(float)atol);
    st->info = (isnan(st->x_norm) || st->residual_norm > st->threshold) ? -1 : 0;
    st->iterations = k;
    st->matvecs = matvecs;
    st->recurrence_rs = rs;
    free(r);
    free(p);
    free(Ap);
    return 0;
}

/* ------------------------------------------------------------------ block-Jacobi
 * z = blockdiag(binv) r: binv holds the inverted bs x bs diagonal blocks row-major per block (a ragged last block completed
 * with identity).  z_i = fma chain over the block's columns in ascending order: restates hipk_block_jacobi_kernel. */
void orc_block_jacobi_apply(int64_t n, int bs, const real *binv, const real *in, real *out) {
#pragma omp parallel for schedule(static) if (g_threads > 1)
    for (int64_t i = 0; i < n; ++i) {
        const int64_t b0 = (i / bs) * bs;
        const real *row = binv + i * bs;
        real acc = (real)0;
        for (int j = 0; j < bs && b0 + j < n; ++j) acc = (real)fma(row[j], in[b0 + j], acc);
        out[i] = acc;
    }
}

/* CG with M = blockdiag(binv) as a CALLABLE between the fused kernels (`_cg_solve` with M, TSL:806-856): restates
 * pytorch_sparse_solver/_hipk.py::solve_cg_stepwise -- z = M r stored, <r,z> and <r,r> plain chunked dots, the rest as
 * orc_pcg_jacobi (with a diagonal M the two agree bit for bit). */
int orc_pcg_blockjacobi(int64_t n, const int32_t *crow, const int32_t *col, const real *val, int bs, const real *binv,
                        const real *b, real *x /* in: x0, out: x */, double tol, double atol, int64_t maxiter,
                        orc_stats *st) {
    memset(st, 0, sizeof(*st));
    if (maxiter < 0) maxiter = 10 * n;
    real *r = (real *)malloc(sizeof(real) * (size_t)n);
    real *p = (real *)malloc(sizeof(real) * (size_t)n);
    real *Ap = (real *)malloc(sizeof(real) * (size_t)n);
    real *z = (real *)malloc(sizeof(real) * (size_t)n);
    const double bs2 = orc_dot(n, b, b);
    const float tolf = (float)tol, atolf = (float)atol;
    const double a2 = (double)(tolf * tolf) * bs2, a3 = (double)(atolf * atolf);
    const double atol2 = a2 > a3 ? a2 : a3;
    orc_spmv(n, crow, col, val, x, b, r);
    int64_t matvecs = 1;
    double rs = orc_dot_tiled(n, r, r); /* fused in the residual SpMV */
    orc_block_jacobi_apply(n, bs, binv, r, z);
    memcpy(p, z, sizeof(real) * (size_t)n);
    double gamma = orc_dot(n, r, z);
    int64_t k = 0;
    while (!(k >= maxiter || rs <= atol2)) {
        orc_spmv(n, crow, col, val, p, NULL, Ap);
        ++matvecs;
        const double pAp = orc_dot_tiled(n, p, Ap);
        const double alpha = gamma / pAp;
#pragma omp parallel for schedule(static) if (g_threads > 1)
        for (int64_t i = 0; i < n; ++i) {
            const real m1 = (real)alpha * Ap[i];
            r[i] = r[i] - m1;
        }
        const double rr = orc_dot(n, r, r);
        orc_block_jacobi_apply(n, bs, binv, r, z);
        const double rz = orc_dot(n, r, z);
        const double beta = rz / gamma;
#pragma omp parallel for schedule(static) if (g_threads > 1)
        for (int64_t i = 0; i < n; ++i) {
            const real m0 = (real)alpha * p[i];
            x[i] = x[i] + m0;
            const real m = (real)beta * p[i];
            p[i] = z[i] + m;
        }
        gamma = rz;
        rs = rr;
        ++k;
    }
    orc_spmv(n, crow, col, val, x, b, Ap);
    ++matvecs;
    orc_block_jacobi_apply(n, bs, binv, Ap, z);
    st->residual_norm = norm_from_sq(orc_dot(n, z, z));
    st->b_norm = norm_from_sq(bs2);
    st->x_norm = norm_from_sq(orc_dot(n, x, x));
    st->threshold = tmax((double)(float)tol * st->b_norm, (double)(float)atol);
    st->info = (isnan(st->x_norm) || st->residual_norm > st->threshold) ? -1 : 0;
    st->iterations = k;
    st->matvecs = matvecs;
    st->recurrence_rs = rs;
    free(r);
    free(p);
    free(Ap);
    free(z);
    return 0;
}

/* ------------------------------------------------------------------ BiCGStab: TSL:859-964 */
/* out = A x (or bsub - A x), then row-scaled by dinv when given: M after A, one extra rounding per element */
static void spmv_m(int64_t n, const int32_t *crow, const int32_t *col, const real *val, const real *dinv, const real *x,
                   const real *bsub, real *out) {
    orc_spmv(n, crow, col, val, x, bsub, out);
    if (dinv != NULL) {
#pragma omp parallel for schedule(static) if (g_threads > 1)
        for (int64_t i = 0; i < n; ++i) out[i] = dinv[i] * out[i];
    }
}


/* dinv != NULL: M = diag(dinv) applied BEFORE A (TSL:908, 922: phat = M p, shat = M s, x += alpha phat + omega shat),
   final `info` from ||M (b - A x)|| (TSL:1007).  phat / shat are stored vectors on the device as well. */
static int bicgstab_impl(int64_t n, const int32_t *crow, const int32_t *col, const real *val, const real *dinv,
                         const real *b, real *x, double tol, double atol, int64_t maxiter, orc_stats *st) {
    csr_t A = {n, crow, col, val};
    memset(st, 0, sizeof(*st));
    if (maxiter < 0) maxiter = 10 * n;
    const size_t nb = sizeof(real) * (size_t)n;
    real *r = (real *)malloc(nb), *rhat = (real *)malloc(nb), *p = (real *)malloc(nb);
    real *q = (real *)malloc(nb), *s = (real *)malloc(nb), *t = (real *)malloc(nb);
    real *phat = dinv ? (real *)malloc(nb) : p, *shat = dinv ? (real *)malloc(nb) : s;
    const double bs = orc_dot(n, b, b);
    const float tolf = (float)tol, atolf = (float)atol;
    const double a2 = (double)(tolf * tolf) * bs, a3 = (double)(atolf * atolf);
    const double atol2 = a2 > a3 ? a2 : a3;
    orc_spmv(n, crow, col, val, x, b, r);
    int64_t matvecs = 1;
    memcpy(rhat, r, nb);
    memcpy(p, r, nb);
    memcpy(q, r, nb);
    double alpha = 1.0, omega = 1.0, rho = 1.0, rs = 0.0;
    int64_t k = 0;
    int code = 0;
    /* <r,r> and <rhat,r> of the NEXT iteration are produced by the x/r update kernel (plain dot spec);
       those of iteration 0 come fused out of the initial residual SpMV (tiled spec), rhat = r0 */
    double rs_next = orc_dot_tiled(n, r, r), rho_next = rs_next;
    while (k < maxiter) {
        rs = rs_next;
        if (rs <= atol2) break;
        const double rho_new = rho_next;
        if (fabs(rho_new) < ORC_EPS * fabs(rho)) {
            code = -10;
            break;
        }
        const double beta = rho_new / rho * alpha / omega; /* left to right, TSL:906 */
#pragma omp parallel for schedule(static) if (g_threads > 1)
        for (int64_t i = 0; i < n; ++i) {
            const real t1 = (real)omega * q[i];
            const real t2 = p[i] - t1;
            const real t3 = (real)beta * t2;
            p[i] = r[i] + t3;
            if (dinv) phat[i] = dinv[i] * p[i];
        }
        orc_spmv(n, crow, col, val, phat, NULL, q);
        ++matvecs;
        const double alpha_new = rho_new / orc_dot_tiled(n, rhat, q);
        if (fabs(alpha_new) < ORC_EPS) {
            code = -11;
            break;
        }
#pragma omp parallel for schedule(static) if (g_threads > 1)
        for (int64_t i = 0; i < n; ++i) {
            const real m = (real)alpha_new * q[i];
            s[i] = r[i] - m;
            if (dinv) shat[i] = dinv[i] * s[i];
        }
        const int exit_early = orc_dot(n, s, s) < atol2;
        orc_spmv(n, crow, col, val, shat, NULL, t);
        ++matvecs;
        const double tt = orc_dot_tiled(n, t, t);
        double omega_new;
        if (fabs(tt) < ORC_EPS)
            omega_new = 0.0;
        else
            omega_new = orc_dot_tiled(n, s, t) / tt;
        if (fabs(omega_new) < ORC_EPS && !exit_early) {
            code = -11;
            break;
        }
#pragma omp parallel for schedule(static) if (g_threads > 1)
        for (int64_t i = 0; i < n; ++i) {
            const real m0 = (real)alpha_new * phat[i];
            if (exit_early) {
                x[i] = x[i] + m0;
                r[i] = s[i];
            } else {
                const real m1 = (real)omega_new * shat[i];
                const real m2 = m0 + m1;
                x[i] = x[i] + m2;
                const real m3 = (real)omega_new * t[i];
                r[i] = s[i] - m3;
            }
        }
        rs_next = orc_dot(n, r, r);
        rho_next = orc_dot(n, rhat, r);
        rho = rho_new;
        alpha = alpha_new;
        omega = omega_new;
        ++k;
        if (exit_early) break;
    }
    if (dinv == NULL) {
        isolve_epilogue(&A, b, x, tol, atol, bs, t, st);
    } else { /* TSL:1007 with M */
        spmv_m(n, crow, col, val, dinv, x, b, t);
        st->residual_norm = norm_from_sq(orc_dot_tiled(n, t, t));
        st->b_norm = norm_from_sq(bs);
        st->x_norm = norm_from_sq(orc_dot(n, x, x));
        st->threshold = tmax((double)(float)tol * st->b_norm, (double)(float)atol);
        st->info = (isnan(st->x_norm) || st->residual_norm > st->threshold) ? -1 : 0;
        free(phat);
        free(shat);
    }
    st->iterations = k;
    st->matvecs = matvecs + 1;
    st->breakdown = code;
    st->recurrence_rs = rs;
    free(r);
    free(rhat);
    free(p);
    free(q);
    free(s);
    free(t);
    return 0;
}

int orc_bicgstab(int64_t n, const int32_t *crow, const int32_t *col, const real *val, const real *b, real *x, double tol,
                 double atol, int64_t maxiter, orc_stats *st) {
    return bicgstab_impl(n, crow, col, val, NULL, b, x, tol, atol, maxiter, st);
}

int orc_bicgstab_jacobi(int64_t n, const int32_t *crow, const int32_t *col, const real *val, const real *dinv,
                        const real *b, real *x, double tol, double atol, int64_t maxiter, orc_stats *st) {
    return bicgstab_impl(n, crow, col, val, dinv, b, x, tol, atol, maxiter, st);
}

/* ------------------------------------------------------------------ GMRES: TSL:641-803, 431-493, 557-638 */
/* `_lstsq` normal equations + Cholesky, fallback general solve (TSL:391-428).
   H is (m+1) x m row-major with leading dimension ldh; uses rows 0..k, cols 0..k-1. */
static void lstsq_normal(const double *H, int ldh, int k, double beta0, double *y) {
    /* restart <= ORC_MAXM: the arrays live on the heap with leading dimension ORC_LD (the arithmetic does not depend on it) */
    double *a2 = (double *)malloc(sizeof(double) * ORC_LD * ORC_LD), *L = (double *)calloc((size_t)ORC_LD * ORC_LD, sizeof(double));
    double *M = (double *)malloc(sizeof(double) * ORC_LD * (ORC_LD + 1));
    double b2[ORC_LD], z[ORC_LD];
    for (int i = 0; i < k; ++i) {
        for (int j = 0; j < k; ++j) {
            double s = 0.0;
            for (int p = 0; p <= k; ++p) s = fma(H[p * ldh + i], H[p * ldh + j], s);
            a2[i * ORC_LD + j] = s;
        }
        b2[i] = H[0 * ldh + i] * beta0; /* beta_vec = [beta0, 0, ...] */
    }
    int ok = 1;
    for (int j = 0; j < k && ok; ++j) {
        double d = a2[j * ORC_LD + j];
        for (int p = 0; p < j; ++p) d = fma(-L[j * ORC_LD + p], L[j * ORC_LD + p], d);
        if (!(d > 0.0)) {
            ok = 0;
            break;
        }
        const double ljj = sqrt(d);
        L[j * ORC_LD + j] = ljj;
        for (int i = j + 1; i < k; ++i) {
            double s = a2[i * ORC_LD + j];
            for (int p = 0; p < j; ++p) s = fma(-L[i * ORC_LD + p], L[j * ORC_LD + p], s);
            L[i * ORC_LD + j] = s / ljj;
        }
    }
    if (ok) {
        for (int i = 0; i < k; ++i) {
            double s = b2[i];
            for (int p = 0; p < i; ++p) s = fma(-L[i * ORC_LD + p], z[p], s);
            z[i] = s / L[i * ORC_LD + i];
        }
        for (int i = k - 1; i >= 0; --i) {
            double s = z[i];
            for (int p = i + 1; p < k; ++p) s = fma(-L[p * ORC_LD + i], y[p], s);
            y[i] = s / L[i * ORC_LD + i];
        }
        free(a2);
        free(L);
        free(M);
        return;
    }
    /* torch.linalg.solve fallback: Gaussian elimination with partial pivoting */
    for (int i = 0; i < k; ++i) {
        for (int j = 0; j < k; ++j) M[i * (ORC_LD + 1) + j] = a2[i * ORC_LD + j];
        M[i * (ORC_LD + 1) + k] = b2[i];
    }
    for (int c = 0; c < k; ++c) {
        int piv = c;
        for (int i = c + 1; i < k; ++i)
            if (fabs(M[i * (ORC_LD + 1) + c]) > fabs(M[piv * (ORC_LD + 1) + c])) piv = i;
        if (piv != c)
            for (int j = 0; j <= k; ++j) {
                const double tmp = M[c * (ORC_LD + 1) + j];
                M[c * (ORC_LD + 1) + j] = M[piv * (ORC_LD + 1) + j];
                M[piv * (ORC_LD + 1) + j] = tmp;
            }
        for (int i = c + 1; i < k; ++i) {
            const double f = M[i * (ORC_LD + 1) + c] / M[c * (ORC_LD + 1) + c];
            for (int j = c; j <= k; ++j) M[i * (ORC_LD + 1) + j] = fma(-f, M[c * (ORC_LD + 1) + j], M[i * (ORC_LD + 1) + j]);
        }
    }
    for (int i = k - 1; i >= 0; --i) {
        double s = M[i * (ORC_LD + 1) + k];
        for (int p = i + 1; p < k; ++p) s = fma(-M[i * (ORC_LD + 1) + p], y[p], s);
        y[i] = s / M[i * (ORC_LD + 1) + i];
    }
    free(a2);
    free(L);
    free(M);
}

/* `_givens_rotation` (TSL:508-518) */
static void givens(double a, double b, double *cs, double *sn) {
    if (fabs(b) == 0.0) {
        *cs = 1.0;
        *sn = 0.0;
        return;
    }
    if (fabs(a) < fabs(b)) {
        const double t = -(a / b);
        const double r = 1.0 / sqrt(1.0 + fabs(t) * fabs(t)); /* torch.rsqrt */
        *cs = r * t;
        *sn = r;
    } else {
        const double t = -(b / a);
        const double r = 1.0 / sqrt(1.0 + fabs(t) * fabs(t));
        *cs = r;
        *sn = r * t;
    }
}

/* gpu_tolerances: 1 = the `device.type == 'cuda'` branch of TSL:737-744 */
/* M = diag(dinv) (left preconditioning: every A(.) of TSL:641-803 is followed by M(.), TSL:351, 791, 766; ptol from
   ||M b||, TSL:750): dinv == NULL is the unpreconditioned solver.  Row scaling is one extra rounding per element, applied
   by the SpMV epilogue on the device (mode bit HIPK_SPMV_SCALE) before the fused dots. */
static int gmres_impl(int64_t n, const int32_t *crow, const int32_t *col, const real *val, const real *dinv,
                      const real *b, real *x, double tol, double atol, int restart, int64_t maxiter,
                      int method /*0 batched,1 incremental*/, int gpu_tolerances, orc_stats *st) {
    memset(st, 0, sizeof(*st));
    if (restart < 1 || restart > ORC_MAXM) return -1;
    if (maxiter < 0) maxiter = 10 * n;
    const int m = restart;
    const size_t nb = sizeof(real) * (size_t)n;
    real *V = (real *)malloc(nb * (size_t)(m + 1)); /* column j at V + j*n */
    real *tmp = (real *)malloc(nb);
    const int ldh = ORC_LD;
    double *H = (double *)malloc(sizeof(double) * (ORC_LD + 1) * ORC_LD), *R = (double *)malloc(sizeof(double) * ORC_LD * ORC_LD);
    double gv[ORC_LD][2], beta_vec[ORC_LD + 1], rvec[ORC_LD], hvec[ORC_LD], y[ORC_LD];

    const double bs = orc_dot(n, b, b);
    const double b_norm = norm_from_sq(bs);
    /* TSL:735-748 */
    const double sq = sqrt((double)n);
    const double cand = (gpu_tolerances ? 1e-12 : 1e-14) * sq;
    const double adaptive = (cand > tol) ? cand : (double)(float)tol; /* python max(): float stays a float -> fp32 tensor */
    const double base_atol = (double)(float)(ORC_EPS64 * (gpu_tolerances ? 1000 : 100) * (double)n);
    const double atol_eff = tmax(adaptive * b_norm, tmax((double)(float)atol, base_atol));
    const double mb_norm = dinv ? norm_from_sq(dot_scaled(n, b, dinv, 0)) : b_norm;  /* ||M b||, TSL:750 */
    const double ptol = mb_norm * tmin(1.0, atol_eff / b_norm);                      /* TSL:750-753 */

    /* TSL:791-792 */
    real *res = V; /* residual lives in column 0 */
    spmv_m(n, crow, col, val, dinv, x, b, res);
    int64_t matvecs = 1;
    double res_norm = norm_from_sq(orc_dot_tiled(n, res, res));
    {
        const int use = res_norm > ORC_EPS;
#pragma omp parallel for schedule(static) if (g_threads > 1)
        for (int64_t i = 0; i < n; ++i) res[i] = use ? res[i] / (real)res_norm : (real)0;
        if (!use) res_norm = 0.0;
    }
    int64_t cycles = 0;
    int happy = 0;
    while (cycles < maxiter && res_norm > atol_eff) {
        memset(H, 0, sizeof(double) * (ORC_LD + 1) * ORC_LD);
        memset(R, 0, sizeof(double) * ORC_LD * ORC_LD);
        for (int i = 0; i < ORC_LD; ++i) R[i * ORC_LD + i] = 1.0; /* TSL:581 */
        memset(gv, 0, sizeof(gv));
        memset(beta_vec, 0, sizeof(beta_vec));
        beta_vec[0] = res_norm;
        int k = 0, breakdown = 0;
        double err = res_norm;
        while (k < m && !breakdown && (method == 0 || err > ptol)) {
            /* ---- `_kth_arnoldi_iteration` (TSL:331-388) */
            real *w = V + (size_t)(k + 1) * n;
            spmv_m(n, crow, col, val, dinv, V + (size_t)k * n, NULL, w);
            ++matvecs;
            double norm0 = norm_from_sq(orc_dot_tiled(n, w, w)); /* fused in the SpMV */
            if (!(norm0 > ORC_EPS)) norm0 = 0.0;
            /* CGS, <= 2 passes (TSL:284-328) */
            for (int j = 0; j <= k; ++j) rvec[j] = 0.0;
            double qnorm = 0.0;
            for (int pass = 0; pass < 2; ++pass) {
                if (pass == 1) {
                    double rr = 0.0;
                    for (int j = 0; j <= k; ++j) rr = fma(rvec[j], rvec[j], rr);
                    double rnorm = norm_from_sq(rr);
                    if (!(rnorm > ORC_EPS)) rnorm = 0.0;
                    if (!(rnorm < qnorm * ORC_INV_SQRT2)) break;
                }
                for (int j = 0; j <= k; ++j) hvec[j] = orc_dot(n, V + (size_t)j * n, w);
#pragma omp parallel for schedule(static) if (g_threads > 1)
                for (int64_t i = 0; i < n; ++i) {
                    double s = 0.0;
                    for (int j = 0; j <= k; ++j) s = fma((double)V[(size_t)j * n + i], hvec[j], s);
                    w[i] = (real)((double)w[i] - s);
                }
                for (int j = 0; j <= k; ++j) rvec[j] = rvec[j] + hvec[j];
                qnorm = norm_from_sq(orc_dot(n, w, w));
                if (!(qnorm > ORC_EPS)) qnorm = 0.0; /* `_safe_normalize(q)` default thresh */
            }
            /* TSL:358-359: thresh = eps * ||A v_k||; the norm is recomputed from q */
            double norm1 = norm_from_sq(orc_dot(n, w, w));
            const double thr = ORC_EPS * norm0;
            const int use = norm1 > thr;
#pragma omp parallel for schedule(static) if (g_threads > 1)
            for (int64_t i = 0; i < n; ++i) w[i] = use ? w[i] / (real)norm1 : (real)0;
            if (!use) norm1 = 0.0;
            for (int j = 0; j <= k; ++j) H[j * ldh + k] = rvec[j];
            H[(k + 1) * ldh + k] = norm1;
            breakdown = (norm1 == 0.0);
            if (method == 1) {
                /* TSL:595-623 */
                double hc[ORC_LD + 1];
                for (int j = 0; j <= k + 1; ++j) hc[j] = H[j * ldh + k];
                for (int i = 0; i < k; ++i) {
                    const double cs = gv[i][0], sn = gv[i][1];
                    const double t0 = cs * hc[i] - sn * hc[i + 1];
                    hc[i + 1] = sn * hc[i] + cs * hc[i + 1];
                    hc[i] = t0;
                }
                double cs, sn;
                givens(hc[k], hc[k + 1], &cs, &sn);
                gv[k][0] = cs;
                gv[k][1] = sn;
                hc[k] = cs * hc[k] - sn * hc[k + 1];
                hc[k + 1] = 0.0;
                for (int j = 0; j <= k; ++j) R[j * ORC_LD + k] = hc[j];
                const double t0 = cs * beta_vec[k] - sn * beta_vec[k + 1];
                beta_vec[k + 1] = sn * beta_vec[k] + cs * beta_vec[k + 1];
                beta_vec[k] = t0;
                err = fabs(beta_vec[k + 1]);
            }
            ++k;
        }
        if (breakdown) happy = 1;
        if (k > 0) {
            if (method == 0) {
                lstsq_normal(H, ldh, k, res_norm, y);
            } else {
                for (int i = k - 1; i >= 0; --i) { /* solve_triangular, TSL:630 */
                    double s = beta_vec[i];
                    for (int p = i + 1; p < k; ++p) s = fma(-R[i * ORC_LD + p], y[p], s);
                    y[i] = s / R[i * ORC_LD + i];
                }
            }
#pragma omp parallel for schedule(static) if (g_threads > 1)
            for (int64_t i = 0; i < n; ++i) {
                double s = 0.0;
                for (int j = 0; j < k; ++j) s = fma((double)V[(size_t)j * n + i], y[j], s);
                x[i] = (real)((double)x[i] + s);
            }
        }
        spmv_m(n, crow, col, val, dinv, x, b, res);
        ++matvecs;
        res_norm = norm_from_sq(orc_dot_tiled(n, res, res));
        {
            const int use = res_norm > ORC_EPS;
#pragma omp parallel for schedule(static) if (g_threads > 1)
            for (int64_t i = 0; i < n; ++i) res[i] = use ? res[i] / (real)res_norm : (real)0;
            if (!use) res_norm = 0.0;
        }
        ++cycles;
    }
    /* TSL:766-773 */
    spmv_m(n, crow, col, val, dinv, x, b, tmp);
    ++matvecs;
    st->residual_norm = norm_from_sq(orc_dot_tiled(n, tmp, tmp));
    st->x_norm = norm_from_sq(orc_dot(n, x, x));
    st->b_norm = b_norm;
    st->threshold = atol_eff * 10;
    st->info = (isnan(st->x_norm) || st->residual_norm > st->threshold) ? -1 : 0;
    st->iterations = cycles;
    st->matvecs = matvecs;
    st->breakdown = happy;
    st->recurrence_rs = res_norm;
    free(V);
    free(tmp);
    free(H);
    free(R);
    return 0;
}

int orc_gmres(int64_t n, const int32_t *crow, const int32_t *col, const real *val, const real *b, real *x, double tol,
              double atol, int restart, int64_t maxiter, int method, int gpu_tolerances, orc_stats *st) {
    return gmres_impl(n, crow, col, val, NULL, b, x, tol, atol, restart, maxiter, method, gpu_tolerances, st);
}

int orc_gmres_jacobi(int64_t n, const int32_t *crow, const int32_t *col, const real *val, const real *dinv, const real *b,
                     real *x, double tol, double atol, int restart, int64_t maxiter, int method, int gpu_tolerances,
                     orc_stats *st) {
    return gmres_impl(n, crow, col, val, dinv, b, x, tol, atol, restart, maxiter, method, gpu_tolerances, st);
}
