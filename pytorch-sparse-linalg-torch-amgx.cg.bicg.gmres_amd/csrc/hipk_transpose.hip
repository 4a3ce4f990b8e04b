// hipk_transpose.hip -- device-side CSR transpose for the adjoint solves of the implicit-diff backward.
//
// The reference's backward solves with A^T (`ImplicitAdjointFunction.backward`, TSL:1237-1248: `A.T` of a dense tensor).
// For a CSR operand the transpose used to go through torch (CSR -> CSC view -> CSR re-conversion); here it is built from
// the handle's own int32 arrays:
//   1. a STABLE radix sort of the entries by column index (keys = col, payload = entry index j in CSR order;
//      hipCUB DeviceRadixSort, stable by construction) -- entries of one column keep their row order, so every row of
//      A^T comes out sorted by column (= source row), the torch CSR invariant the bit-exact row sums rely on;
//   2. row pointers of A^T = lower bounds of 0..n_cols in the sorted keys;
//   3. col_t[k] = source row of entry perm[k] (upper bound in crow), val_t[k] = val[perm[k]].
// A setup step (once per matrix, cached on the handle): 1-2 ms at nnz = 20 M.  Not on the per-iteration path.
#include <hipcub/hipcub.hpp>

#include "hipk_common.h"
#include "hipk_solve.h"

__global__ void hipk_iota_kernel(int *__restrict__ p, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = (int)i;
}

// crow_t[c] = first position k with keys[k] >= c   (keys sorted ascending), c = 0..n_cols
__global__ void hipk_lower_bounds_kernel(const int *__restrict__ keys, int64_t nnz, int64_t n_cols, int *__restrict__ crow_t) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c <= n_cols; c += stride) {
        int64_t lo = 0, hi = nnz;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (keys[mid] < (int)c) lo = mid + 1; else hi = mid;
        }
        crow_t[c] = (int)lo;
    }
}

// entry k of A^T: column = the source row of entry perm[k] (last r with crow[r] <= j), value = val[perm[k]]
template <typename T>
__global__ void hipk_transpose_fill_kernel(const int *__restrict__ crow, int64_t n_rows, const int *__restrict__ perm,
                                           const T *__restrict__ val, int64_t nnz, int *__restrict__ col_t,
                                           T *__restrict__ val_t) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += stride) {
        const int j = perm[k];
        int64_t lo = 0, hi = n_rows;  // invariant: crow[lo] <= j < crow[hi]
        while (hi - lo > 1) {
            const int64_t mid = (lo + hi) >> 1;
            if (crow[mid] <= j) lo = mid; else hi = mid;
        }
        col_t[k] = (int)lo;
        val_t[k] = val[j];
    }
}

static int hipk_key_bits(int64_t n_cols) {
    int bits = 1;
    while (bits < 31 && ((int64_t)1 << bits) < n_cols) ++bits;
    return bits;
}

static size_t hipk_sort_temp_bytes(int64_t nnz, int end_bit) {
    size_t tmp = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, tmp, (const int *)nullptr, (int *)nullptr, (const int *)nullptr,
                                             (int *)nullptr, (int)nnz, 0, end_bit, (hipStream_t)0);
    return tmp;
}

extern "C" size_t hipk_csr_transpose_work_bytes(hipk_csr_t h) {
    if (!h) return 0;
    const size_t arr = hipk_align_up(sizeof(int) * (size_t)(h->nnz > 0 ? h->nnz : 1), 256);
    return 3 * arr + hipk_align_up(hipk_sort_temp_bytes(h->nnz, hipk_key_bits(h->n_cols)), 256) + 256;
}

extern "C" int hipk_csr_transpose(hipk_csr_t h, int32_t *crow_t, int32_t *col_t, void *val_t, void *work, size_t work_bytes,
                                  hipk_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    HIPK_REQUIRE(h && crow_t && work, HIPK_ERR_ARG, "null argument");
    HIPK_REQUIRE(h->nnz == 0 || (col_t && val_t), HIPK_ERR_ARG, "col_t / val_t is null");
    HIPK_REQUIRE(work_bytes >= hipk_csr_transpose_work_bytes(h), HIPK_ERR_WORKSPACE, "work too small");
    HIPK_REQUIRE((((uintptr_t)work) & 255u) == 0, HIPK_ERR_ALIGN, "work must be 256-byte aligned");
    const int64_t nnz = h->nnz, n_cols = h->n_cols;
    if (nnz == 0) {
        HIPK_CHECK_HIP(hipMemsetAsync(crow_t, 0, sizeof(int) * (size_t)(n_cols + 1), stream));
        return HIPK_OK;
    }
    const size_t arr = hipk_align_up(sizeof(int) * (size_t)nnz, 256);
    char *w = (char *)work;
    int *perm_in = (int *)w, *keys_out = (int *)(w + arr), *perm_out = (int *)(w + 2 * arr);
    void *tmp = w + 3 * arr;
    const int end_bit = hipk_key_bits(n_cols);
    size_t tmp_bytes = hipk_sort_temp_bytes(nnz, end_bit);
    int grid = (int)((nnz + 255) / 256);
    if (grid > 8192) grid = 8192;
    hipk_iota_kernel<<<grid, 256, 0, stream>>>(perm_in, nnz);
    HIPK_CHECK_HIP(hipGetLastError());
    HIPK_CHECK_HIP(hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, (const int *)h->col, keys_out, (const int *)perm_in, perm_out,
                                                      (int)nnz, 0, end_bit, stream));
    int cgrid = (int)((n_cols + 1 + 255) / 256);
    if (cgrid > 8192) cgrid = 8192;
    hipk_lower_bounds_kernel<<<cgrid, 256, 0, stream>>>(keys_out, nnz, n_cols, crow_t);
    if (h->dtype == HIPK_F64)
        hipk_transpose_fill_kernel<double><<<grid, 256, 0, stream>>>(h->crow, h->n_rows, perm_out, (const double *)h->val, nnz,
                                                                     col_t, (double *)val_t);
    else
        hipk_transpose_fill_kernel<float><<<grid, 256, 0, stream>>>(h->crow, h->n_rows, perm_out, (const float *)h->val, nnz,
                                                                    col_t, (float *)val_t);
    HIPK_CHECK_HIP(hipGetLastError());
    return HIPK_OK;
}
