// hipk_blas1.hip -- deterministic dot and the two-kernel axpy replacements.
//   hipk_dot   <- `_vdot_real_tree`  (TSL:130-139)
//   hipk_axpy  <- `_add(y, _mul(a, x))` (TSL:847): mul then add, two roundings
//   hipk_xpby  <- `_add(x, _mul(b, y))` (TSL:852)
#include "hipk_blas1.h"

template <typename T>
__global__ __launch_bounds__(HIPK_THREADS) void hipk_dot_kernel(int64_t n, int ch, const T *__restrict__ x,
                                                                const T *__restrict__ y,
                                                                double *__restrict__ part) {
    __shared__ double sbuf[HIPK_THREADS];
    const int c = blockIdx.x;
    double acc = 0.0;
    hipk_chunk_loop<T>(n, ch, c, [&](int64_t i, int nv) {
        T xv[hipk_vec<T>::VEC], yv[hipk_vec<T>::VEC];
        hipk_ld<T>(x, i, nv, xv);
        hipk_ld<T>(y, i, nv, yv);
#pragma unroll
        for (int k = 0; k < hipk_vec<T>::VEC; ++k)
            if (k < nv) acc = fma((double)xv[k], (double)yv[k], acc);
    });
    acc = hipk_block_sum(acc, sbuf);
    if (threadIdx.x == 0) part[c] = acc;
}

__global__ __launch_bounds__(HIPK_THREADS) void hipk_finish1_kernel(const double *__restrict__ part, int g,
                                                                    double *__restrict__ out) {
    __shared__ double sbuf[HIPK_THREADS];
    const double r = hipk_reduce_parts(part, g, sbuf);
    if (threadIdx.x == 0) out[0] = r;
}

template <typename T, int OP>
__global__ __launch_bounds__(HIPK_THREADS) void hipk_axpy_kernel(int64_t n, int ch, double a_,
                                                                 const T *__restrict__ x, T *__restrict__ y) {
    const int c = blockIdx.x;
    const T a = (T)a_;
    hipk_chunk_loop<T>(n, ch, c, [&](int64_t i, int nv) {
        T xv[hipk_vec<T>::VEC], yv[hipk_vec<T>::VEC];
        hipk_ld<T>(x, i, nv, xv);
        hipk_ld<T>((const T *)y, i, nv, yv);
#pragma unroll
        for (int k = 0; k < hipk_vec<T>::VEC; ++k) {
            if (OP == 0) {
                const T m = a * xv[k];  // y = y + a*x
                yv[k] = yv[k] + m;
            } else {
                const T m = a * yv[k];  // y = x + a*y
                yv[k] = xv[k] + m;
            }
        }
        hipk_st<T>(y, i, nv, yv);
    });
}

int hipk_launch_finish1(const double *part, int g, double *out_dev, hipStream_t stream) {
    hipk_finish1_kernel<<<1, HIPK_THREADS, 0, stream>>>(part, g, out_dev);
    HIPK_CHECK_HIP(hipGetLastError());
    return HIPK_OK;
}

int hipk_launch_dot_parts(int64_t n, const void *x, const void *y, int dtype, double *part,
                          hipStream_t stream) {
    return hipk_launch_dot_parts_ch(n, hipk_make_geom(n).ch, x, y, dtype, part, stream);
}

int hipk_launch_dot_parts_ch(int64_t n, int ch, const void *x, const void *y, int dtype, double *part,
                             hipStream_t stream) {
    hipk_geom gm;
    gm.n = n;
    gm.ch = ch;
    gm.g = (int)((n + ch - 1) / ch);
    if (n <= 0) return HIPK_OK;
    if (dtype == HIPK_F64)
        hipk_dot_kernel<double><<<gm.g, HIPK_THREADS, 0, stream>>>(n, gm.ch, (const double *)x,
                                                                   (const double *)y, part);
    else
        hipk_dot_kernel<float><<<gm.g, HIPK_THREADS, 0, stream>>>(n, gm.ch, (const float *)x,
                                                                  (const float *)y, part);
    HIPK_CHECK_HIP(hipGetLastError());
    return HIPK_OK;
}

extern "C" int hipk_dot(int64_t n, const void *x, const void *y, int dtype, double *out_dev,
                        void *scratch_dev, hipk_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    HIPK_REQUIRE(n >= 0 && x && y && out_dev && scratch_dev, HIPK_ERR_ARG, "bad argument");
    HIPK_REQUIRE(dtype == HIPK_F64 || dtype == HIPK_F32, HIPK_ERR_UNSUPPORTED, "dtype");
    HIPK_REQUIRE(hipk_aligned16(x) && hipk_aligned16(y), HIPK_ERR_ALIGN, "x/y must be 16-byte aligned");
    int rc = hipk_launch_dot_parts(n, x, y, dtype, (double *)scratch_dev, stream);
    if (rc != HIPK_OK) return rc;
    return hipk_launch_finish1((const double *)scratch_dev, n > 0 ? hipk_make_geom(n).g : 0, out_dev, stream);
}

static int hipk_axpy_like(int op, int64_t n, double a, const void *x, void *y, int dtype, hipStream_t stream) {
    HIPK_REQUIRE(n >= 0 && x && y, HIPK_ERR_ARG, "bad argument");
    HIPK_REQUIRE(dtype == HIPK_F64 || dtype == HIPK_F32, HIPK_ERR_UNSUPPORTED, "dtype");
    HIPK_REQUIRE(hipk_aligned16(x) && hipk_aligned16(y), HIPK_ERR_ALIGN, "x/y must be 16-byte aligned");
    if (n == 0) return HIPK_OK;
    const hipk_geom gm = hipk_make_geom(n);
    if (dtype == HIPK_F64) {
        if (op == 0)
            hipk_axpy_kernel<double, 0><<<gm.g, HIPK_THREADS, 0, stream>>>(n, gm.ch, a, (const double *)x, (double *)y);
        else
            hipk_axpy_kernel<double, 1><<<gm.g, HIPK_THREADS, 0, stream>>>(n, gm.ch, a, (const double *)x, (double *)y);
    } else {
        if (op == 0)
            hipk_axpy_kernel<float, 0><<<gm.g, HIPK_THREADS, 0, stream>>>(n, gm.ch, a, (const float *)x, (float *)y);
        else
            hipk_axpy_kernel<float, 1><<<gm.g, HIPK_THREADS, 0, stream>>>(n, gm.ch, a, (const float *)x, (float *)y);
    }
    HIPK_CHECK_HIP(hipGetLastError());
    return HIPK_OK;
}

extern "C" int hipk_axpy(int64_t n, double a, const void *x, void *y, int dtype, hipk_stream_t stream) {
    return hipk_axpy_like(0, n, a, x, y, dtype, (hipStream_t)stream);
}
extern "C" int hipk_xpby(int64_t n, const void *x, double b, void *y, int dtype, hipk_stream_t stream) {
    return hipk_axpy_like(1, n, b, x, y, dtype, (hipStream_t)stream);
}

extern "C" int hipk_dot_parts(int64_t n, int chunk_rows, const void *x, const void *y, int dtype, double *part,
                              hipk_stream_t stream) {
    HIPK_REQUIRE(n >= 0 && x && y && part, HIPK_ERR_ARG, "bad argument");
    HIPK_REQUIRE(dtype == HIPK_F64 || dtype == HIPK_F32, HIPK_ERR_UNSUPPORTED, "dtype");
    HIPK_REQUIRE(chunk_rows >= HIPK_BASE_CHUNK && chunk_rows % HIPK_BASE_CHUNK == 0, HIPK_ERR_ARG, "chunk_rows");
    HIPK_REQUIRE((n + chunk_rows - 1) / chunk_rows <= HIPK_MAX_PARTS, HIPK_ERR_ARG, "too many chunks");
    HIPK_REQUIRE(hipk_aligned16(x) && hipk_aligned16(y), HIPK_ERR_ALIGN, "x/y must be 16-byte aligned");
    return hipk_launch_dot_parts_ch(n, chunk_rows, x, y, dtype, part, (hipStream_t)stream);
}

template <typename T>
__global__ void hipk_gather_kernel(int64_t m, const int *__restrict__ idx, const T *__restrict__ src,
                                   T *__restrict__ dst) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += stride) dst[i] = src[idx[i]];
}

extern "C" int hipk_gather(int64_t m, const int32_t *idx_dev, const void *src, void *dst, int dtype,
                           hipk_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    HIPK_REQUIRE(m >= 0, HIPK_ERR_ARG, "negative size");
    if (m == 0) return HIPK_OK;
    HIPK_REQUIRE(idx_dev && src && dst, HIPK_ERR_ARG, "null argument");
    HIPK_REQUIRE(dtype == HIPK_F64 || dtype == HIPK_F32, HIPK_ERR_UNSUPPORTED, "dtype");
    int grid = (int)((m + 255) / 256);
    if (grid > 2048) grid = 2048;
    if (dtype == HIPK_F64)
        hipk_gather_kernel<double><<<grid, 256, 0, stream>>>(m, idx_dev, (const double *)src, (double *)dst);
    else
        hipk_gather_kernel<float><<<grid, 256, 0, stream>>>(m, idx_dev, (const float *)src, (float *)dst);
    HIPK_CHECK_HIP(hipGetLastError());
    return HIPK_OK;
}

extern "C" int hipk_reduce_parts(const double *part_dev, int g, double *out_dev, hipk_stream_t stream) {
    HIPK_REQUIRE(part_dev && out_dev && g >= 0 && g <= HIPK_MAX_PARTS, HIPK_ERR_ARG, "bad argument");
    return hipk_launch_finish1(part_dev, g, out_dev, (hipStream_t)stream);
}

// ---------------------------------------------------------------- block-Jacobi preconditioner
// z = M r with M = blockdiag(A)^-1: the inverses of the bs x bs diagonal blocks, row-major per block (binv[b][i][j]),
// precomputed once.  The reference's hook is any callable `M` (TSL:849, 908, 922, 351); this is the device kernel a
// `BlockJacobiPreconditioner` runs between the fused solver kernels.  Thread per row: z_i = fma-chain over the block's
// columns in ascending order (mirrored by oracle/krylov_oracle.c::orc_block_jacobi_apply): bitwise reproducible.
// Traffic: bs*sizeof(T) per row of binv (streamed, non-temporal) + the vector in and out.
template <typename T, int BS>
__global__ __launch_bounds__(HIPK_THREADS) void hipk_block_jacobi_kernel(int64_t n, int bs_rt, const T *__restrict__ binv,
                                                                         const T *__restrict__ in, T *__restrict__ out) {
    const int bs = BS > 0 ? BS : bs_rt;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int64_t b0 = (i / bs) * bs;
        const T *__restrict__ row = binv + i * bs;   // block b, local row i - b0: (b*bs + (i - b0)) * bs = i * bs
        T acc = (T)0;
#pragma unroll
        for (int j = 0; j < (BS > 0 ? BS : 32); ++j) {
            if (j < bs && b0 + j < n) acc = fma(__builtin_nontemporal_load(row + j), in[b0 + j], acc);
        }
        out[i] = acc;
    }
}

extern "C" int hipk_block_jacobi_apply(int64_t n, int block_size, const void *binv_dev, const void *in, void *out, int dtype,
                                       hipk_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    HIPK_REQUIRE(n >= 0 && block_size >= 1 && block_size <= 32, HIPK_ERR_ARG, "block_size must be in [1, 32]");
    if (n == 0) return HIPK_OK;
    HIPK_REQUIRE(binv_dev && in && out, HIPK_ERR_ARG, "null argument");
    HIPK_REQUIRE(in != out, HIPK_ERR_ARG, "in and out must not alias");
    HIPK_REQUIRE(dtype == HIPK_F64 || dtype == HIPK_F32, HIPK_ERR_UNSUPPORTED, "dtype");
    int grid = (int)((n + HIPK_THREADS - 1) / HIPK_THREADS);
    if (grid > 16384) grid = 16384;
#define HIPK_BJ(T, B) hipk_block_jacobi_kernel<T, B><<<grid, HIPK_THREADS, 0, stream>>>(n, block_size, (const T *)binv_dev, (const T *)in, (T *)out)
#define HIPK_BJ_T(T)                          \
    switch (block_size) {                     \
        case 2: HIPK_BJ(T, 2); break;         \
        case 4: HIPK_BJ(T, 4); break;         \
        case 8: HIPK_BJ(T, 8); break;         \
        case 16: HIPK_BJ(T, 16); break;       \
        default: HIPK_BJ(T, 0); break;        \
    }
    if (dtype == HIPK_F64) {
        HIPK_BJ_T(double)
    } else {
        HIPK_BJ_T(float)
    }
#undef HIPK_BJ_T
#undef HIPK_BJ
    HIPK_CHECK_HIP(hipGetLastError());
    return HIPK_OK;
}
