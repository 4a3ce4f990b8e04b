// How fast can the SpMV of a constant-coefficient stencil tile go when nothing uniform across the tile is re-derived per lane?
// y_i = sum_k val_k x[i + off_k] (entries ascending, multiply then add: the row-sum spec), fused <x, y>, one reduction chunk
// (8 tiles of 256 rows) per workgroup, the library's wavefront sum and chunk fold.  Variants:
//   0: offsets / values as kernel arguments (scalar registers), all entries present
//   1: + a 64-bit presence mask per (tile, entry, wavefront), scalar-loaded, applied as the EXEC mask
//   2: as 1, but the masks are only read for tiles flagged "ragged" (scalar branch per tile)
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I include -I <pkg>/csrc tools/ubench/stencil_probe.hip -o tools/ubench/stencil_probe
#include "hipk_spmv.h"
#include "hipk_coded.h"

#include <vector>

#define CK(x)                                                                        \
    do {                                                                             \
        hipError_t e_ = (x);                                                         \
        if (e_ != hipSuccess) {                                                      \
            printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__);    \
            return 1;                                                                \
        }                                                                            \
    } while (0)

struct pat_t {
    long long boff[5];  // byte offsets
    double val[5];
};

template <int VAR>
__global__ __launch_bounds__(HIPK_THREADS) void stencil_kernel(int n, int g, pat_t pt, const char *__restrict__ xb,
                                                               double *__restrict__ y, const double *__restrict__ w,
                                                               const unsigned long long *__restrict__ masks,
                                                               const unsigned char *__restrict__ ragged,
                                                               double *__restrict__ part0) {
    constexpr int TPC = 8;
    const int chunk = hipk_xcd_chunk(blockIdx.x, g);
    if (chunk < 0) return;
    __shared__ double wsum0[TPC * 4];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int t_first = chunk * TPC;
    const int ntiles = (n + 255) >> 8;
    const int t_end = t_first + TPC < ntiles ? t_first + TPC : ntiles;
    auto tile = [&](int tl, double(&xv)[5], double &wv, unsigned long long(&m)[5]) {
        const unsigned vo = (unsigned)(tl * 256 + t) * 8u;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            m[k] = ~0ull;
            if (VAR == 1 || (VAR == 2 && ragged[tl])) m[k] = masks[((size_t)tl * 5 + k) * 4 + wave];
            xv[k] = 0.0;
            if (VAR == 0) {
                xv[k] = *(const double *)(xb + pt.boff[k] + vo);
            } else if (__builtin_amdgcn_inverse_ballot_w64(m[k])) {
                xv[k] = *(const double *)(xb + pt.boff[k] + vo);
            }
        }
        wv = w[tl * 256 + t];
    };
    auto finish = [&](int tl, const double(&xv)[5], double wv, const unsigned long long(&m)[5]) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            if (VAR == 0) {
                s = s + pt.val[k] * xv[k];
            } else if (__builtin_amdgcn_inverse_ballot_w64(m[k])) {
                s = s + pt.val[k] * xv[k];
            }
        }
        y[tl * 256 + t] = s;
        double d0 = hipk_wave_sum(wv * s);
        if (lane == 0) wsum0[(tl - t_first) * 4 + wave] = d0;
    };
    for (int tp = t_first; tp < t_end; tp += 2) {
        double xa[5], xc[5], wa, wc;
        unsigned long long ma[5], mc[5];
        tile(tp, xa, wa, ma);
        if (tp + 1 < t_end) tile(tp + 1, xc, wc, mc);
        finish(tp, xa, wa, ma);
        if (tp + 1 < t_end) finish(tp + 1, xc, wc, mc);
    }
    __syncthreads();
    if (t < 64) {
        const double r = hipk_wave_fold(wsum0, t_end - t_first, lane);
        if (lane == 0) part0[chunk] = r;
    }
}


// lean form only: TPT tiles per loop trip (all their loads in flight together); SHARE = 1: x[i-1], x[i+1] and w come from
// the x[i] load (DPP lane shifts; the two edge lanes of a wavefront take theirs with a one-lane load) -- three
// 512-byte requests per tile and wavefront instead of six
template <int TPT, int SHARE>
__global__ __launch_bounds__(HIPK_THREADS) void stencil_lean_kernel(int n, int g, pat_t pt, const char *__restrict__ xb,
                                                                    double *__restrict__ y, const double *__restrict__ w,
                                                                    double *__restrict__ part0) {
    constexpr int TPC = 8;
    const int chunk = hipk_xcd_chunk(blockIdx.x, g);
    if (chunk < 0) return;
    __shared__ double wsum0[TPC * 4];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int t_first = chunk * TPC;
    const int ntiles = (n + 255) >> 8;
    const int t_end = t_first + TPC < ntiles ? t_first + TPC : ntiles;
    for (int tp = t_first; tp < t_end; tp += TPT) {
        double xv[TPT][5], wv[TPT], ed[TPT];
#pragma unroll
        for (int q = 0; q < TPT; ++q) {
            if (tp + q < t_end) {
                const unsigned vo = (unsigned)((tp + q) * 256 + t) * 8u;
                if (SHARE) {
                    xv[q][0] = *(const double *)(xb + pt.boff[0] + vo);
                    xv[q][2] = *(const double *)(xb + pt.boff[2] + vo);
                    xv[q][4] = *(const double *)(xb + pt.boff[4] + vo);
                    ed[q] = 0.0;
                    if (lane == 0) ed[q] = *(const double *)(xb + pt.boff[1] + vo);
                    if (lane == 63) ed[q] = *(const double *)(xb + pt.boff[3] + vo);
                } else {
#pragma unroll
                    for (int k = 0; k < 5; ++k) xv[q][k] = *(const double *)(xb + pt.boff[k] + vo);
                    wv[q] = w[(tp + q) * 256 + t];
                }
            }
        }
#pragma unroll
        for (int q = 0; q < TPT; ++q) {
            if (tp + q < t_end) {
                if (SHARE) {
                    const double c = xv[q][2];
                    double up = __shfl_up(c, 1), dn = __shfl_down(c, 1);
                    xv[q][1] = lane == 0 ? ed[q] : up;
                    xv[q][3] = lane == 63 ? ed[q] : dn;
                    wv[q] = c;
                }
                double s = 0.0;
#pragma unroll
                for (int k = 0; k < 5; ++k) s = s + pt.val[k] * xv[q][k];
                y[(tp + q) * 256 + t] = s;
                double d0 = hipk_wave_sum(wv[q] * s);
                if (lane == 0) wsum0[(tp + q - t_first) * 4 + wave] = d0;
            }
        }
    }
    __syncthreads();
    if (t < 64) {
        const double r = hipk_wave_fold(wsum0, t_end - t_first, lane);
        if (lane == 0) part0[chunk] = r;
    }
}


// decomposition of the lean kernel's time: LOADS = 1 (x[i] only), 3 (+ x[i -+ nx]), 5 (all five, separate loads); DOT = fused
// dot on/off; W2 = 1: two adjacent rows per lane (16-byte accesses; no dot)
template <int LOADS, int DOT, int W2>
__global__ __launch_bounds__(HIPK_THREADS) void stencil_parts_kernel(int n, int g, pat_t pt, const char *__restrict__ xb,
                                                                     double *__restrict__ y, double *__restrict__ part0) {
    constexpr int TPC = 8;
    const int chunk = hipk_xcd_chunk(blockIdx.x, g);
    if (chunk < 0) return;
    __shared__ double wsum0[TPC * 4];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int t_first = chunk * TPC;
    const int ntiles = (n + 255) >> 8;
    const int t_end = t_first + TPC < ntiles ? t_first + TPC : ntiles;
    if (W2) {
        for (int tp = t_first; tp < t_end; tp += 2) {   // 512 rows per trip, two per lane
            const unsigned vo = (unsigned)(tp * 256 + 2 * t) * 8u;
            double2 xv[5];
#pragma unroll
            for (int k = 0; k < 5; ++k)
                if (k == 2 || (LOADS >= 3 && (k == 0 || k == 4)) || LOADS == 5) xv[k] = *(const double2 *)(xb + pt.boff[k] + vo);
            double2 s = {0.0, 0.0};
#pragma unroll
            for (int k = 0; k < 5; ++k)
                if (k == 2 || (LOADS >= 3 && (k == 0 || k == 4)) || LOADS == 5) {
                    s.x = s.x + pt.val[k] * xv[k].x;
                    s.y = s.y + pt.val[k] * xv[k].y;
                }
            *(double2 *)((char *)y + vo) = s;
        }
        return;
    }
    for (int tp = t_first; tp < t_end; tp += 2) {
        double xv[2][5];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const unsigned vo = (unsigned)((tp + q) * 256 + t) * 8u;
#pragma unroll
            for (int k = 0; k < 5; ++k)
                if (k == 2 || (LOADS >= 3 && (k == 0 || k == 4)) || LOADS == 5) xv[q][k] = *(const double *)(xb + pt.boff[k] + vo);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < 5; ++k)
                if (k == 2 || (LOADS >= 3 && (k == 0 || k == 4)) || LOADS == 5) s = s + pt.val[k] * xv[q][k];
            y[(tp + q) * 256 + t] = s;
            if (DOT) {
                double d0 = hipk_wave_sum(xv[q][2] * s);
                if (lane == 0) wsum0[(tp + q - t_first) * 4 + wave] = d0;
            }
        }
    }
    if (DOT) {
        __syncthreads();
        if (t < 64) {
            const double r = hipk_wave_fold(wsum0, t_end - t_first, lane);
            if (lane == 0) part0[chunk] = r;
        }
    }
}


// memory shape of a fused "p = r + beta p, x += alpha p, q = A p" pass (two rows per lane): r and p at the five offsets, x;
// stores p_new (other buffer), x, q; fused <p_new, q>.  FUSED = 0: the same two passes as separate loops in one kernel is not
// meaningful -- compare with the sum of the stand-alone kernels instead.
__global__ __launch_bounds__(HIPK_THREADS) void stencil_fused_kernel(int n, int g, pat_t pt, const char *__restrict__ rb,
                                                                     const char *__restrict__ pb, double *__restrict__ xv_,
                                                                     double *__restrict__ pn, double *__restrict__ q,
                                                                     double alpha, double beta, double *__restrict__ part0) {
    constexpr int TPC = 8;
    const int chunk = hipk_xcd_chunk(blockIdx.x, g);
    if (chunk < 0) return;
    __shared__ double wsum0[TPC * 4];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wp = wave >> 1, wh = wave & 1;
    const int t_first = chunk * TPC;
    const int ntiles = (n + 255) >> 8;
    const int t_end = t_first + TPC < ntiles ? t_first + TPC : ntiles;
    for (int tl = t_first + wp; tl < t_end; tl += 2) {
        const unsigned vo = (unsigned)(tl * 256 + wh * 128 + 2 * lane) * 8u;
        double2 rv[5], pv[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            rv[k] = *(const double2 *)(rb + pt.boff[k] + vo);
            pv[k] = *(const double2 *)(pb + pt.boff[k] + vo);
        }
        double2 xx = *(const double2 *)((const char *)xv_ + vo);
        double2 s = {0.0, 0.0}, pc = {0.0, 0.0};
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            double2 pj;
            pj.x = rv[k].x + beta * pv[k].x;
            pj.y = rv[k].y + beta * pv[k].y;
            if (k == 2) pc = pj;
            s.x = s.x + pt.val[k] * pj.x;
            s.y = s.y + pt.val[k] * pj.y;
        }
        xx.x = xx.x + alpha * pv[2].x;
        xx.y = xx.y + alpha * pv[2].y;
        *(double2 *)((char *)xv_ + vo) = xx;
        *(double2 *)((char *)pn + vo) = pc;
        *(double2 *)((char *)q + vo) = s;
        double2 d = {pc.x * s.x, pc.y * s.y};
        const double r = hipk_half_tree2(d);
        if ((lane & 31) == 0) wsum0[(tl - t_first) * 4 + 2 * wh + (lane >> 5)] = r;
    }
    __syncthreads();
    if (t < 64) {
        const double r = hipk_wave_fold(wsum0, t_end - t_first, lane);
        if (lane == 0) part0[chunk] = r;
    }
}

int main(int argc, char **argv) {
    const int nx = argc > 1 ? atoi(argv[1]) : 2000;
    const int n = nx * nx, ntiles = (n + 255) / 256, g = (ntiles + 7) / 8;
    const size_t padn = (size_t)n + 2 * (size_t)nx + 512;
    double *x, *y, *part;
    unsigned long long *masks;
    unsigned char *ragged;
    CK(hipMalloc(&x, padn * 8));
    CK(hipMalloc(&y, (size_t)ntiles * 256 * 8));
    CK(hipMalloc(&part, 4096 * 8));
    CK(hipMalloc(&masks, (size_t)ntiles * 5 * 4 * 8));
    CK(hipMalloc(&ragged, ntiles));
    std::vector<double> hx(padn, 0.0);
    for (int i = 0; i < n; ++i) hx[nx + 256 + i] = 1.0 + 1e-3 * (i % 97);
    CK(hipMemcpy(x, hx.data(), padn * 8, hipMemcpyHostToDevice));
    // 2-D 5-point Laplacian, row = ix * nx + iy: entries -nx, -1, 0, +1, +nx where the neighbour exists
    std::vector<unsigned long long> hm((size_t)ntiles * 20, 0ull);
    std::vector<unsigned char> hr(ntiles, 0);
    const int offs[5] = {-nx, -1, 0, 1, nx};
    for (int i = 0; i < n; ++i) {
        const int ix = i / nx, iy = i % nx;
        const bool pres[5] = {ix > 0, iy > 0, true, iy < nx - 1, ix < nx - 1};
        const int tl = i >> 8, wv = (i & 255) >> 6, ln = i & 63;
        for (int k = 0; k < 5; ++k)
            if (pres[k]) hm[((size_t)tl * 5 + k) * 4 + wv] |= 1ull << ln;
    }
    int n_ragged = 0;
    for (int tl = 0; tl < ntiles; ++tl) {
        bool full = true;
        for (int q = 0; q < 20; ++q) full = full && hm[(size_t)tl * 20 + q] == ~0ull;
        hr[tl] = full ? 0 : 1;
        n_ragged += hr[tl];
    }
    CK(hipMemcpy(masks, hm.data(), hm.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(ragged, hr.data(), ntiles, hipMemcpyHostToDevice));
    pat_t pt;
    for (int k = 0; k < 5; ++k) {
        pt.boff[k] = (long long)offs[k] * 8;
        pt.val[k] = k == 2 ? 4.0 : -1.0;
    }
    const char *xb = (const char *)(x + nx + 256);
    const double *w = x + nx + 256;
    const int grid = hipk_xcd_grid(g);
    printf("nx %d n %d tiles %d (ragged %d) chunks %d grid %d\n", nx, n, ntiles, n_ragged, g, grid);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<double> ref;
    for (int var = 0; var < 3; ++var) {
        const int reps = 400;
        for (int it = 0; it < reps + 20; ++it) {
            if (it == 20) CK(hipEventRecord(e0, 0));
            if (var == 0) stencil_kernel<0><<<grid, HIPK_THREADS>>>(n, g, pt, xb, y, w, masks, ragged, part);
            if (var == 1) stencil_kernel<1><<<grid, HIPK_THREADS>>>(n, g, pt, xb, y, w, masks, ragged, part);
            if (var == 2) stencil_kernel<2><<<grid, HIPK_THREADS>>>(n, g, pt, xb, y, w, masks, ragged, part);
        }
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<double> hy(n);
        CK(hipMemcpy(hy.data(), y, (size_t)n * 8, hipMemcpyDeviceToHost));
        double chk = 0;
        for (int i = 0; i < n; ++i) chk += hy[i];
        if (var == 1) ref = hy;
        bool same = true;
        if (var == 2) same = memcmp(ref.data(), hy.data(), (size_t)n * 8) == 0;
        printf("variant %d: %.2f us per launch (back to back), sum(y) %.6e%s\n", var, 1e3 * ms / reps, chk,
               var == 2 ? (same ? "  == variant 1" : "  DIFFERS from variant 1") : "");
    }
    for (int var = 0; var < 8; ++var) {
        const int reps = 400;
        for (int it = 0; it < reps + 20; ++it) {
            if (it == 20) CK(hipEventRecord(e0, 0));
#define LK(T, S) stencil_lean_kernel<T, S><<<grid, HIPK_THREADS>>>(n, g, pt, xb, y, w, part)
            if (var == 0) LK(1, 0);
            if (var == 1) LK(2, 0);
            if (var == 2) LK(4, 0);
            if (var == 3) LK(8, 0);
            if (var == 4) LK(1, 1);
            if (var == 5) LK(2, 1);
            if (var == 6) LK(4, 1);
            if (var == 7) LK(8, 1);
        }
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<double> hy(n);
        CK(hipMemcpy(hy.data(), y, (size_t)n * 8, hipMemcpyDeviceToHost));
        double chk = 0;
        for (int i = 0; i < n; ++i) chk += hy[i];
        printf("lean: %d tiles per trip, share %d: %.2f us per launch, sum(y) %.6e\n", 1 << (var & 3), var >> 2, 1e3 * ms / reps, chk);
    }
    for (int var = 0; var < 10; ++var) {
        const int reps = 400;
        const char *names[10] = {"1 load, no dot", "1 load, dot", "3 loads, no dot", "3 loads, dot", "5 loads, no dot", "5 loads, dot",
                                 "1 load, 16 B per lane", "3 loads, 16 B per lane", "5 loads, 16 B per lane", "-"};
        if (var == 9) break;
        for (int it = 0; it < reps + 20; ++it) {
            if (it == 20) CK(hipEventRecord(e0, 0));
#define PK(L, D, W) stencil_parts_kernel<L, D, W><<<grid, HIPK_THREADS>>>(n, g, pt, xb, y, part)
            if (var == 0) PK(1, 0, 0);
            if (var == 1) PK(1, 1, 0);
            if (var == 2) PK(3, 0, 0);
            if (var == 3) PK(3, 1, 0);
            if (var == 4) PK(5, 0, 0);
            if (var == 5) PK(5, 1, 0);
            if (var == 6) PK(1, 0, 1);
            if (var == 7) PK(3, 0, 1);
            if (var == 8) PK(5, 0, 1);
        }
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("parts: %-24s %.2f us per launch\n", names[var], 1e3 * ms / reps);
    }
    {
        double *rr_, *pp_, *pn_, *q_, *x2_;
        CK(hipMalloc(&rr_, padn * 8));
        CK(hipMalloc(&pp_, padn * 8));
        CK(hipMalloc(&pn_, padn * 8));
        CK(hipMalloc(&q_, padn * 8));
        CK(hipMalloc(&x2_, padn * 8));
        CK(hipMemcpy(rr_, hx.data(), padn * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(pp_, hx.data(), padn * 8, hipMemcpyHostToDevice));
        CK(hipMemset(x2_, 0, padn * 8));
        const int reps = 400;
        for (int it = 0; it < reps + 20; ++it) {
            if (it == 20) CK(hipEventRecord(e0, 0));
            stencil_fused_kernel<<<grid, HIPK_THREADS>>>(n, g, pt, (const char *)(rr_ + nx + 256), (const char *)(pp_ + nx + 256),
                                                        x2_ + nx + 256, pn_ + nx + 256, q_ + nx + 256, 1e-3, 0.5, part);
        }
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("fused direction + SpMV shape (r, p at five offsets, x; stores p', x, q; 192 MB): %.2f us per launch\n", 1e3 * ms / reps);
    }
    return 0;
}
