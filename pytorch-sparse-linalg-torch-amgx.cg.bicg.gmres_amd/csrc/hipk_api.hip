// hipk_api.hip -- handle management, error plumbing and the SpMV entry points.
#include <stdarg.h>

#include "hipk_blas1.h"
#include "hipk_common.h"
#include "hipk_solve.h"
#include "hipk_spmv.h"
#include "hipk_coded.h"
#include <stdlib.h>

#include <vector>

// ------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";

void hipk_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *hipk_last_error(void) { return g_err; }
static thread_local char g_spmv_kernel[96] = "";
extern "C" const char *hipk_last_spmv_kernel(void) { return g_spmv_kernel; }
#define HIPK_NOTE_KERNEL(...) snprintf(g_spmv_kernel, sizeof(g_spmv_kernel), __VA_ARGS__)
#if __has_include("hipk_build_id.h")   // written by the Makefile (sha1 over the sources); absent in ad-hoc compiles of this file
#include "hipk_build_id.h"
#else
#define HIPK_BUILD_ID "unstamped-build!"
#endif
extern "C" int hipk_version(void) { return HIPK_VERSION; }
extern "C" const char *hipk_build_id(void) { return HIPK_BUILD_ID; }

extern "C" int hipk_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    int ok = 0;
    for (int d = 0; d < n; ++d) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, d) == hipSuccess && strncmp(p.gcnArchName, "gfx950", 6) == 0) ++ok;
    }
    return ok;
}

extern "C" int hipk_chunk_size(int64_t n) { return hipk_make_geom(n).ch; }
extern "C" int hipk_chunk_count(int64_t n) { return hipk_make_geom(n).g; }
extern "C" size_t hipk_scratch_bytes(void) {
    return (size_t)HIPK_SCRATCH_SLOTS * HIPK_MAX_PARTS * sizeof(double);
}

// ------------------------------------------------------------------ CSR handle
// Narrow torch's int64 indices to int32 and validate the structure on the way, so a
// malformed matrix is rejected here instead of faulting inside the SpMV gather.
template <typename I>
__global__ void hipk_narrow_check_kernel(const I *__restrict__ crow_in, const I *__restrict__ col_in,
                                         int *__restrict__ crow, int *__restrict__ col, int64_t n_rows,
                                         int64_t n_cols, int64_t nnz, int *__restrict__ bad) {
    // bad[0]: error flags; bad[1]: longest row; bad[2]: most entries in a 256-row tile.
    // Flags and maxima are combined per wavefront first: one atomic per wavefront instead of one per row
    // (4 M atomics on one address made this kernel 20x slower than its memory traffic).
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int flags = 0, max_len = 0, max_tile = 0;
    for (int64_t i = i0; i <= n_rows; i += stride) {
        const int64_t v = (int64_t)crow_in[i];
        if (v < 0 || v > nnz) flags |= 1;
        if (i == 0 && v != 0) flags |= 2;
        if (i == n_rows && v != nnz) flags |= 4;
        if (i < n_rows && (int64_t)crow_in[i + 1] < v) flags |= 8;
        if (i < n_rows) {
            const int64_t len = (int64_t)crow_in[i + 1] - v;
            const int l32 = (int)(len > INT32_MAX ? INT32_MAX : (len < 0 ? 0 : len));
            max_len = l32 > max_len ? l32 : max_len;
            if ((i & 255) == 0) {
                const int64_t e = (i + 256 < n_rows) ? i + 256 : n_rows;
                const int64_t tl = (int64_t)crow_in[e] - v;
                const int t32 = (int)(tl > INT32_MAX ? INT32_MAX : (tl < 0 ? 0 : tl));
                max_tile = t32 > max_tile ? t32 : max_tile;
            }
        }
        crow[i] = (int)v;
    }
    for (int64_t j = i0; j < nnz; j += stride) {
        const int64_t v = (int64_t)col_in[j];
        if (v < 0 || v >= n_cols) flags |= 16;
        col[j] = (int)v;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        flags |= __shfl_down(flags, o);
        const int a = __shfl_down(max_len, o), b = __shfl_down(max_tile, o);
        max_len = a > max_len ? a : max_len;
        max_tile = b > max_tile ? b : max_tile;
    }
    if ((threadIdx.x & 63) == 0) {
        if (flags) atomicOr(bad, flags);
        if (max_len > 0) atomicMax(bad + 1, max_len);
        if (max_tile > 0) atomicMax(bad + 2, max_tile);
    }
}

// rows with more than `cap` entries -> list (order irrelevant: each row is independent)
__global__ void hipk_huge_rows_kernel(const int *__restrict__ crow, int64_t n_rows, int cap, int *__restrict__ list,
                                      int *__restrict__ count) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_rows; r += stride)
        if (crow[r + 1] - crow[r] > cap) list[atomicAdd(count, 1)] = (int)r;
}


// Try to build the coded form (hipk_coded.h).  Failure of any kind just leaves the handle on the plain kernels.
// OFFS_ONLY: dictionary of column offsets only + per-entry value planes (variable-coefficient stencils; layout 3).
template <typename T, bool OFFS_ONLY>
static hipError_t hipk_build_coded(hipk_csr_s *h, hipStream_t stream) {
    hipk_dict_table *tb = nullptr;
    hipError_t e = hipMalloc((void **)&tb, sizeof(hipk_dict_table));
    if (e != hipSuccess) return e;
    struct guard {
        hipk_dict_table *p;
        ~guard() { (void)hipFree(p); }
    } g{tb};
    e = hipMemsetAsync(tb, 0, sizeof(hipk_dict_table), stream);
    if (e != hipSuccess) return e;
    int grid = (int)((h->n_rows + 255) / 256);
    if (grid > 8192) grid = 8192;
    if (!OFFS_ONLY && h->n_rows > 8192) {
        // sample first: the pairs of the leading 4096 rows.  A matrix with per-entry values overflows the dictionary
        // here, in a few microseconds, instead of in a full pass with thousands of threads fighting over the table.
        hipk_dict_insert_kernel<T, OFFS_ONLY><<<16, HIPK_THREADS, 0, stream>>>(h->crow, h->col, (const T *)h->val, 4096, tb);
        int head[3] = {0, 0, 0};  // count, overflow, fail
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(head, &tb->count, sizeof(head), hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        if (e != hipSuccess) return e;
        if (head[1] || head[0] > HIPK_CODED_MAX) return hipSuccess;
    }
    hipk_dict_insert_kernel<T, OFFS_ONLY><<<grid, HIPK_THREADS, 0, stream>>>(h->crow, h->col, (const T *)h->val, h->n_rows,
                                                                            tb);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipk_dict_table *ht = new hipk_dict_table;
    struct hguard {
        hipk_dict_table *p;
        ~hguard() { delete p; }
    } hg{ht};
    e = hipMemcpyAsync(ht, tb, sizeof(hipk_dict_table), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return e;
    if (ht->overflow || ht->count > HIPK_CODED_MAX || ht->count <= 0) return hipSuccess;
    // number the occupied slots; the order is irrelevant to the results (a lookup returns the exact pair)
    int off_h[HIPK_CODED_MAX];
    T val_h[HIPK_CODED_MAX];
    memset(off_h, 0, sizeof(off_h));
    memset(val_h, 0, sizeof(val_h));
    int nc = 0;
    for (int s = 0; s < HIPK_DICT_SLOTS; ++s) {
        ht->slot_code[s] = 0;
        if (ht->key[s] == 0) continue;
        if (nc >= HIPK_CODED_MAX) return hipSuccess;
        ht->slot_code[s] = nc;
        off_h[nc] = ht->off[s];
        if (sizeof(T) == 8) {
            memcpy(&val_h[nc], &ht->bits[s], 8);
        } else {
            const unsigned int b32 = (unsigned int)ht->bits[s];
            memcpy(&val_h[nc], &b32, 4);
        }
        ++nc;
    }
    if (nc != ht->count) return hipSuccess;  // a claimed slot whose count was not yet added cannot happen after the sync
    e = hipMemcpyAsync(tb->slot_code, ht->slot_code, sizeof(ht->slot_code), hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipMalloc((void **)&h->dict_off, sizeof(int) * HIPK_CODED_MAX);
    if (e == hipSuccess) e = hipMalloc((void **)&h->dict_val, sizeof(T) * HIPK_CODED_MAX);
    if (e == hipSuccess) e = hipMemcpyAsync(h->dict_off, off_h, sizeof(off_h), hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipMemcpyAsync(h->dict_val, val_h, sizeof(val_h), hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) return e;

    // layout of the code bytes: sliced-ELL planes when the padding stays small, else CSR order + row lengths
    const char *lay = getenv("HIPK_SPMV_CODED_LAYOUT");
    // (the persistent sliced-ELL kernel needs n_rows <= n_cols and 32-bit byte offsets into x)
    const bool want_sell = (OFFS_ONLY || !(lay && strcmp(lay, "csr") == 0)) && nc <= HIPK_SELL_PAD &&
                           h->n_rows <= h->n_cols && (uint64_t)h->n_cols * sizeof(T) < (1ull << 32);
    if (OFFS_ONLY && !want_sell) return hipSuccess;  // the offset-coded form exists in the sliced-ELL layout only
    const int ntiles = (int)((h->n_rows + HIPK_TILE - 1) / HIPK_TILE);
    std::vector<int> toff;
    bool sell = false;
    if (want_sell) {
        int *tw = nullptr;
        e = hipMalloc((void **)&tw, sizeof(int) * (size_t)(ntiles + 1));
        if (e != hipSuccess) return e;
        hipk_tile_width_kernel<<<ntiles, HIPK_THREADS, 0, stream>>>(h->crow, h->n_rows, tw);
        toff.resize((size_t)ntiles + 1);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(toff.data(), tw, sizeof(int) * (size_t)ntiles, hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        if (e != hipSuccess) {
            (void)hipFree(tw);
            return e;
        }
        int64_t planes = 0;
        int wmin = INT32_MAX, wmax = 0;
        for (int i = 0; i < ntiles; ++i) {
            const int w = hipk_sell_units(toff[i]);  // tile size in units of 256 B
            wmin = w < wmin ? w : wmin;
            wmax = w > wmax ? w : wmax;
            toff[i] = (int)planes;
            planes += w;
        }
        toff[ntiles] = (int)planes;
        // nearly uniform sizes (a stencil's first/last grid line is narrower): pad every tile to the largest,
        // so that a tile is found without the offset table (one dependent load less per workgroup)
        const int64_t planes_u = (int64_t)ntiles * wmax;
        if (wmin != wmax && planes_u <= planes + planes / 50 + 8) {
            for (int i = 0; i <= ntiles; ++i) toff[i] = i * wmax;
            planes = planes_u;
            wmin = wmax;
        }
        sell = planes * HIPK_TILE <= 2 * h->nnz + 65536 && planes < (int64_t)INT32_MAX / HIPK_TILE * 64;
        if (sell) {
            h->tile_off = tw;
            h->sell_w = (wmin == wmax) ? wmax : 0;
            h->sell_bytes = planes * HIPK_TILE;
            e = hipMemcpyAsync(tw, toff.data(), sizeof(int) * (size_t)(ntiles + 1), hipMemcpyHostToDevice, stream);
            if (e == hipSuccess) e = hipMalloc((void **)&h->code, (size_t)h->sell_bytes + 256);
            if (e == hipSuccess) e = hipMemsetAsync(h->code, HIPK_SELL_PAD, (size_t)h->sell_bytes + 256, stream);
            if (OFFS_ONLY) {
                if (e == hipSuccess) e = hipMalloc(&h->sell_vals, ((size_t)h->sell_bytes + 256) * sizeof(T));
                if (e == hipSuccess) e = hipMemsetAsync(h->sell_vals, 0, ((size_t)h->sell_bytes + 256) * sizeof(T), stream);
            }
            if (e == hipSuccess) {
                hipk_dict_encode_sell_kernel<T, OFFS_ONLY><<<grid, HIPK_THREADS, 0, stream>>>(
                    h->crow, h->col, (const T *)h->val, h->n_rows, tb, h->tile_off, h->code, (T *)h->sell_vals);
                e = hipGetLastError();
            }
            // tiles whose rows all carry the same code bytes (constant-coefficient stencils: all but the grid-line ends)
            const char *uenv = getenv("HIPK_SPMV_UNIFORM");
            if (e == hipSuccess && !(uenv && uenv[0] == '0')) {
                int *ucount = nullptr;
                e = hipMalloc((void **)&h->tile_ucode, sizeof(unsigned long long) * (size_t)ntiles + 2 * sizeof(int));
                if (e == hipSuccess) {
                    ucount = (int *)(h->tile_ucode + ntiles);
                    e = hipMemsetAsync(ucount, 0, 2 * sizeof(int), stream);
                }
                if (e == hipSuccess) {
                    hipk_tile_uniform_kernel<<<(ntiles + 15) / 16, HIPK_THREADS, 0, stream>>>(h->code, h->tile_off, ntiles, h->tile_ucode,
                                                                                  ucount);
                    e = hipGetLastError();
                }
                if (e == hipSuccess)
                    e = hipMemcpyAsync(&h->n_uniform_tiles, ucount, sizeof(int), hipMemcpyDeviceToHost, stream);
                if (e == hipSuccess)
                    e = hipMemcpyAsync(&h->uniform_units, ucount + 1, sizeof(int), hipMemcpyDeviceToHost, stream);
                // masked tiles (hipk_tile_masked_kernel): pair codes in fp64 only (the two-rows-per-lane kernel's domain), code 254
                // free for the marker.  OPT-IN (HIPK_SPMV_MASKED=1): bit-identical, but measured SLOWER where it was meant to help
                // (N = 4 M: 15.7-16.2 vs 14.6-14.9 us stand-alone, CG 17.6 vs 17.97 k it/s; N = 1.96 M: equal;
                // profiles/r03_spmv_wide_stamps.md section 4)
                const char *menv = getenv("HIPK_SPMV_MASKED");
                if (e == hipSuccess && !OFFS_ONLY && sizeof(T) == 8 && nc <= 254 && menv && menv[0] == '1') {
                    int *mcount = nullptr;
                    e = hipMalloc((void **)&h->tile_wcode, sizeof(unsigned long long) * (size_t)ntiles + 2 * sizeof(int));
                    if (e == hipSuccess) e = hipMalloc((void **)&h->row_mask, (size_t)ntiles * HIPK_TILE + 16);
                    if (e == hipSuccess) {
                        mcount = (int *)(h->tile_wcode + ntiles);
                        e = hipMemsetAsync(mcount, 0, 2 * sizeof(int), stream);
                    }
                    if (e == hipSuccess) {
                        hipk_tile_masked_kernel<<<(ntiles + 15) / 16, HIPK_THREADS, 0, stream>>>(h->code, h->tile_off, ntiles, h->n_rows,
                                                                                     h->dict_off, h->tile_ucode, h->tile_wcode,
                                                                                     h->row_mask, mcount);
                        e = hipGetLastError();
                    }
                    if (e == hipSuccess) e = hipMemcpyAsync(&h->n_masked_tiles, mcount, sizeof(int), hipMemcpyDeviceToHost, stream);
                    if (e == hipSuccess) e = hipMemcpyAsync(&h->masked_units, mcount + 1, sizeof(int), hipMemcpyDeviceToHost, stream);
                }
            }
        } else {
            (void)hipFree(tw);
        }
    }
    if (OFFS_ONLY && !sell) return e;  // planes mostly padding: no offset-coded form
    if (!sell) {
        if (e == hipSuccess) e = hipMalloc((void **)&h->code, (size_t)h->nnz + 32);
        if (e == hipSuccess) e = hipMalloc((void **)&h->rowlen, (size_t)h->n_rows + 16);
        if (e == hipSuccess) e = hipMemsetAsync(h->code + h->nnz, 0, 32, stream);
        if (e == hipSuccess) {
            hipk_dict_encode_kernel<T><<<grid, HIPK_THREADS, 0, stream>>>(h->crow, h->col, (const T *)h->val, h->n_rows,
                                                                         tb, h->code, h->rowlen);
            e = hipGetLastError();
        }
    }
    int fail = 1;
    if (e == hipSuccess) e = hipMemcpyAsync(&fail, &tb->fail, sizeof(int), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);  // also keeps the host staging arrays alive until the copies are done
    if (e != hipSuccess) return e;
    if (!fail) {
        h->n_codes = nc;
        h->coded_layout = OFFS_ONLY ? 3 : (sell ? 2 : 1);
    }
    if (h->tile_ucode && (fail || 4 * (int64_t)h->n_uniform_tiles < ntiles)) {  // too few uniform tiles to pay for the test
        (void)hipFree(h->tile_ucode);
        h->tile_ucode = nullptr;
        h->n_uniform_tiles = 0;
        h->uniform_units = 0;
    }
    if (h->tile_wcode && (fail || !h->tile_ucode || h->n_masked_tiles == 0)) {   // nothing gained: the kernels read tile_ucode
        (void)hipFree(h->tile_wcode);
        if (h->row_mask) (void)hipFree(h->row_mask);
        h->tile_wcode = nullptr;
        h->row_mask = nullptr;
        h->n_masked_tiles = 0;
        h->masked_units = 0;
    }
    return hipSuccess;
}

static void hipk_drop_coded(hipk_csr_s *h) {
    if (h->code) (void)hipFree(h->code);
    if (h->rowlen) (void)hipFree(h->rowlen);
    if (h->dict_off) (void)hipFree(h->dict_off);
    if (h->dict_val) (void)hipFree(h->dict_val);
    if (h->tile_off) (void)hipFree(h->tile_off);
    if (h->tile_ucode) (void)hipFree(h->tile_ucode);
    h->tile_ucode = nullptr;
    h->n_uniform_tiles = 0;
    h->uniform_units = 0;
    if (h->tile_wcode) (void)hipFree(h->tile_wcode);
    if (h->row_mask) (void)hipFree(h->row_mask);
    h->tile_wcode = nullptr;
    h->row_mask = nullptr;
    h->n_masked_tiles = 0;
    h->masked_units = 0;
    if (h->sell_vals) (void)hipFree(h->sell_vals);
    h->sell_vals = nullptr;
    h->tile_off = nullptr;
    h->coded_layout = 0;
    h->code = h->rowlen = nullptr;
    h->dict_off = nullptr;
    h->dict_val = nullptr;
    h->n_codes = 0;
}

extern "C" int hipk_csr_create(hipk_csr_t *out, int64_t n_rows, int64_t n_cols, int64_t nnz,
                               const void *crow_dev, const void *col_dev, int idx_bytes,
                               const void *val_dev, int dtype, hipk_stream_t stream_) {
    return hipk_csr_create_ex(out, n_rows, n_cols, nnz, crow_dev, col_dev, idx_bytes, val_dev, dtype, 0, stream_);
}

extern "C" int hipk_csr_create_ex(hipk_csr_t *out, int64_t n_rows, int64_t n_cols, int64_t nnz,
                                  const void *crow_dev, const void *col_dev, int idx_bytes,
                                  const void *val_dev, int dtype, int chunk_rows, hipk_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (chunk_rows != 0) {
        bool ok = chunk_rows >= HIPK_BASE_CHUNK && (chunk_rows % HIPK_BASE_CHUNK) == 0;
        const int q = ok ? chunk_rows / HIPK_BASE_CHUNK : 0;
        ok = ok && (q & (q - 1)) == 0 && (n_rows + chunk_rows - 1) / chunk_rows <= HIPK_MAX_PARTS;
        HIPK_REQUIRE(ok, HIPK_ERR_ARG, "chunk_rows must be 2048*2^k and give at most 2048 chunks");
    }
    HIPK_REQUIRE(out != nullptr, HIPK_ERR_ARG, "out is null");
    *out = nullptr;
    HIPK_REQUIRE(n_rows >= 0 && n_cols >= 0 && nnz >= 0, HIPK_ERR_ARG, "negative size");
    HIPK_REQUIRE(n_rows < INT32_MAX && n_cols < INT32_MAX && nnz < INT32_MAX, HIPK_ERR_UNSUPPORTED,
                 "rows/cols/nnz must fit int32 (indices are narrowed to 32 bit)");
    HIPK_REQUIRE(idx_bytes == 4 || idx_bytes == 8, HIPK_ERR_ARG, "idx_bytes must be 4 or 8");
    HIPK_REQUIRE(dtype == HIPK_F64 || dtype == HIPK_F32, HIPK_ERR_UNSUPPORTED, "dtype must be f32/f64");
    HIPK_REQUIRE(crow_dev != nullptr, HIPK_ERR_ARG, "crow is null");
    HIPK_REQUIRE(nnz == 0 || (col_dev != nullptr && val_dev != nullptr), HIPK_ERR_ARG, "col/val is null");
    HIPK_REQUIRE(hipk_aligned16(val_dev), HIPK_ERR_ALIGN, "val must be 16-byte aligned");

    hipk_csr_s *h = new hipk_csr_s();
    memset(h, 0, sizeof(*h));
    h->n_rows = n_rows;
    h->n_cols = n_cols;
    h->nnz = nnz;
    h->dtype = dtype;
    h->val = val_dev;
    h->geom = hipk_make_geom(n_rows);
    if (chunk_rows != 0) {  // row block of a larger problem: use the GLOBAL chunk size
        h->geom.ch = chunk_rows;
        h->geom.g = (int)((n_rows + chunk_rows - 1) / chunk_rows);
        if (h->geom.g < 1) h->geom.g = 1;
    }
    int *bad = nullptr;
    int bad3[3] = {0, 0, 0};
    int &bad_h = bad3[0];
    hipError_t e = hipGetDevice(&h->device);
    if (e == hipSuccess) e = hipMalloc((void **)&h->crow, sizeof(int) * (size_t)(n_rows + 1));
    if (e == hipSuccess) e = hipMalloc((void **)&h->col, sizeof(int) * (size_t)(nnz > 0 ? nnz : 1));
    if (e == hipSuccess) e = hipMalloc((void **)&bad, 3 * sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void **)&h->tile_part, sizeof(double) * 8 * (size_t)((n_rows + 255) / 256 + 1));
    if (e == hipSuccess) e = hipHostMalloc((void **)&h->host_poll, 16 * sizeof(int64_t), hipHostMallocDefault);
    if (e == hipSuccess) e = hipMemsetAsync(bad, 0, 3 * sizeof(int), stream);
    if (e == hipSuccess) {
        const int64_t work = (nnz > n_rows + 1) ? nnz : n_rows + 1;
        int grid = (int)((work + 255) / 256);
        if (grid > 4096) grid = 4096;
        if (grid < 1) grid = 1;
        if (idx_bytes == 8)
            hipk_narrow_check_kernel<int64_t><<<grid, 256, 0, stream>>>((const int64_t *)crow_dev,
                                                                       (const int64_t *)col_dev, h->crow,
                                                                       h->col, n_rows, n_cols, nnz, bad);
        else
            hipk_narrow_check_kernel<int><<<grid, 256, 0, stream>>>((const int *)crow_dev, (const int *)col_dev,
                                                                   h->crow, h->col, n_rows, n_cols, nnz, bad);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(bad3, bad, 3 * sizeof(int), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (bad) (void)hipFree(bad);
    if (e != hipSuccess || bad_h != 0) {
        if (e != hipSuccess)
            hipk_set_error("hipk_csr_create: %s", hipGetErrorString(e));
        else
            hipk_set_error("hipk_csr_create: malformed CSR structure (flags 0x%x: 1 crow range, 2 crow[0]!=0, "
                           "4 crow[n]!=nnz, 8 crow not monotone, 16 column out of range)", bad_h);
        hipk_csr_destroy(h);
        return e != hipSuccess ? HIPK_ERR_HIP : HIPK_ERR_ARG;
    }
    h->max_row_len = bad3[1];
    h->max_tile_nnz = bad3[2];
    // short-rowed matrix with a few rows beyond the LDS product buffer: list them for the row-per-wavefront pre-pass
    const int cap = (dtype == HIPK_F64) ? 1280 : 2048;
    if (n_rows > 0 && h->max_row_len > cap && nnz / n_rows < 48) {
        const int64_t max_huge = nnz / cap + 1;
        int *cnt = nullptr;
        int cnt_h = 0;
        e = hipMalloc((void **)&h->huge_rows, sizeof(int) * (size_t)max_huge);
        if (e == hipSuccess) e = hipMalloc((void **)&cnt, sizeof(int));
        if (e == hipSuccess) e = hipMemsetAsync(cnt, 0, sizeof(int), stream);
        if (e == hipSuccess) {
            int grid = (int)((n_rows + 255) / 256);
            if (grid > 4096) grid = 4096;
            hipk_huge_rows_kernel<<<grid, 256, 0, stream>>>(h->crow, n_rows, cap, h->huge_rows, cnt);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(&cnt_h, cnt, sizeof(int), hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        if (cnt) (void)hipFree(cnt);
        if (e != hipSuccess) {
            hipk_set_error("hipk_csr_create (huge rows): %s", hipGetErrorString(e));
            hipk_csr_destroy(h);
            return HIPK_ERR_HIP;
        }
        h->n_huge = cnt_h;
    }
    // coded form: short rows everywhere, mean row below the row-per-wavefront threshold, not disabled by the environment
    h->sell_chunked = 2;  // chunk-per-workgroup form when sell_chunked * chunks >= resident workgroups (0: never)
    if (const char *sc = getenv("HIPK_SPMV_SELL_CHUNKED")) h->sell_chunked = atoi(sc) < 0 ? 0 : (atoi(sc) == 1 ? 2 : atoi(sc));
    h->sell_loop = 1;  // persistent sliced-ELL kernel: grid = sell_loop x the resident workgroups
    if (const char *sl = getenv("HIPK_SPMV_SELL_LOOP")) h->sell_loop = atoi(sl) < 1 ? 1 : (atoi(sl) > 4 ? 4 : atoi(sl));
    {
        hipDeviceProp_t prop;
        h->n_cu = (hipGetDeviceProperties(&prop, h->device) == hipSuccess && prop.multiProcessorCount > 0)
                      ? prop.multiProcessorCount : 256;
    }
    const char *env = getenv("HIPK_SPMV_CODED");
    if (n_rows > 0 && nnz > 0 && h->max_row_len <= HIPK_LONG_ROW && !(env && env[0] == '0')) {
        e = (dtype == HIPK_F64) ? hipk_build_coded<double, false>(h, stream) : hipk_build_coded<float, false>(h, stream);
        if (e != hipSuccess) (void)hipGetLastError();  // e.g. out of memory: stay on the plain kernels
        if (h->n_codes == 0) {
            // too many distinct (offset, value) pairs: try offsets alone, values kept per entry (9 B instead of 12 B per
            // entry, no row pointers) -- HIPK_SPMV_OFFSET_CODED=0 skips it
            hipk_drop_coded(h);
            const char *oc = getenv("HIPK_SPMV_OFFSET_CODED");
            if (!(oc && oc[0] == '0')) {
                e = (dtype == HIPK_F64) ? hipk_build_coded<double, true>(h, stream) : hipk_build_coded<float, true>(h, stream);
                if (e != hipSuccess) (void)hipGetLastError();
            }
            if (h->n_codes == 0) hipk_drop_coded(h);
        }
    }
    *out = h;
    return HIPK_OK;
}

extern "C" int hipk_op_create(hipk_csr_t *out, int64_t n, int dtype, hipk_op_fn op, void *user, hipk_stream_t stream_) {
    (void)stream_;
    HIPK_REQUIRE(out != nullptr, HIPK_ERR_ARG, "out is null");
    *out = nullptr;
    HIPK_REQUIRE(op != nullptr, HIPK_ERR_ARG, "operator callback is null");
    HIPK_REQUIRE(n > 0 && n < INT32_MAX, HIPK_ERR_ARG, "n out of range");
    HIPK_REQUIRE(dtype == HIPK_F64 || dtype == HIPK_F32, HIPK_ERR_UNSUPPORTED, "dtype must be f32/f64");
    hipk_csr_s *h = new hipk_csr_s();
    memset(h, 0, sizeof(*h));
    h->n_rows = h->n_cols = n;
    h->dtype = dtype;
    h->geom = hipk_make_geom(n);
    h->op_cb = op;
    h->op_user = user;
    h->max_row_len = INT32_MAX;   // nothing that holds matrix rows in registers / LDS applies
    h->max_tile_nnz = INT32_MAX;
    hipError_t e = hipGetDevice(&h->device);
    if (e == hipSuccess) e = hipMalloc((void **)&h->tile_part, sizeof(double) * 8 * (size_t)((n + 255) / 256 + 1));
    if (e == hipSuccess) e = hipHostMalloc((void **)&h->host_poll, 16 * sizeof(int64_t), hipHostMallocDefault);
    {
        hipDeviceProp_t prop;
        h->n_cu = (hipGetDeviceProperties(&prop, h->device) == hipSuccess && prop.multiProcessorCount > 0)
                      ? prop.multiProcessorCount : 256;
    }
    if (e != hipSuccess) {
        hipk_set_error("hipk_op_create: %s", hipGetErrorString(e));
        hipk_csr_destroy(h);
        return HIPK_ERR_HIP;
    }
    *out = h;
    return HIPK_OK;
}

// epilogue of a matrix-free product (y already holds A x): residual form, row scaling, the SpMV kernels' fused dots as
// per-wavefront tile sums (the "tiled dot" of the spec, as hipk_rowdot_kernel forms them for the row-per-wavefront path)
template <typename T>
__global__ __launch_bounds__(HIPK_THREADS) void hipk_op_epilogue_kernel(hipk_spmv_args a) {
    if (a.stop_it != nullptr && a.it >= *a.stop_it) return;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int tile = blockIdx.x;
    const int64_t r = (int64_t)tile * HIPK_TILE + t;
    double d0 = 0.0, d1 = 0.0;
    if (r < a.n) {
        T out = ((const T *)a.y)[r];
        if (a.mode & HIPK_SPMV_RESID) out = ((const T *)a.bsub)[r] - out;
        if (a.mode & HIPK_SPMV_SCALE) out = ((const T *)a.dscale)[r] * out;
        if (a.mode & (HIPK_SPMV_RESID | HIPK_SPMV_SCALE)) ((T *)a.y)[r] = out;
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

extern "C" int hipk_csr_destroy(hipk_csr_t h) {
    if (!h) return HIPK_OK;
    if (h->crow) (void)hipFree(h->crow);
    if (h->col) (void)hipFree(h->col);
    if (h->tile_part) (void)hipFree(h->tile_part);
    if (h->huge_rows) (void)hipFree(h->huge_rows);
    if (h->mid_plan_mem) (void)hipFree(h->mid_plan_mem);
    hipk_drop_coded(h);
    if (h->host_poll) (void)hipHostFree(h->host_poll);
    delete h;
    return HIPK_OK;
}

extern "C" int64_t hipk_csr_rows(hipk_csr_t h) { return h ? h->n_rows : -1; }
extern "C" int64_t hipk_csr_nnz(hipk_csr_t h) { return h ? h->nnz : -1; }
extern "C" int64_t hipk_csr_spmv_bytes(hipk_csr_t h) {
    if (!h) return -1;
    const int64_t sv = (h->dtype == HIPK_F64) ? 8 : 4;
    return h->nnz * (sv + 4) + (h->n_rows + 1) * 4 + 2 * h->n_rows * sv;
}

extern "C" int hipk_csr_spmv_path(hipk_csr_t h) {
    if (!h) return -1;
    if (h->n_rows > 0 && h->nnz / h->n_rows >= 48) return HIPK_PATH_ROWWAVE;
    if (h->n_codes > 0 && h->path_override != 1) return h->coded_layout == 3 ? HIPK_PATH_OFFSET_CODED : HIPK_PATH_CODED;
    return (h->max_tile_nnz <= 2048 && h->max_row_len <= HIPK_LONG_ROW) ? HIPK_PATH_TILE_FAST : HIPK_PATH_TILE;
}
extern "C" int hipk_csr_set_path(hipk_csr_t h, int mode) {
    HIPK_REQUIRE(h != nullptr, HIPK_ERR_ARG, "null handle");
    HIPK_REQUIRE(mode == 0 || mode == 1, HIPK_ERR_ARG, "mode must be 0 (auto) or 1 (plain CSR kernels only)");
    h->path_override = mode;
    h->n_plans = 0;
    return HIPK_OK;
}
extern "C" int64_t hipk_csr_format_bytes(hipk_csr_t h) {
    if (!h) return -1;
    const int64_t sv = (h->dtype == HIPK_F64) ? 8 : 4;
    // tiles whose rows share their code bytes: one 8-byte word per tile is read instead of their code planes
    const int64_t ntl = (h->n_rows + 255) / 256;
    // uniform tiles: one 8-byte word instead of their planes; masked tiles (two-rows-per-lane kernel): the word + 256 mask bytes
    const int64_t uni = (h->tile_ucode ? ntl * 8 - (int64_t)h->uniform_units * HIPK_TILE : 0) +
                        (h->tile_wcode ? (int64_t)h->n_masked_tiles * HIPK_TILE - (int64_t)h->masked_units * HIPK_TILE : 0);
    if (hipk_csr_spmv_path(h) == HIPK_PATH_OFFSET_CODED)  // code + value planes (+ plane offsets unless uniform) + x + y
        return h->sell_bytes * (1 + sv) + uni + (h->sell_w > 0 ? 0 : ntl * 8) + 2 * h->n_rows * sv;
    if (hipk_csr_spmv_path(h) == HIPK_PATH_CODED) {
        if (h->coded_layout == 2)  // byte planes (+ two plane offsets per tile unless uniform) + x + y
            return h->sell_bytes + uni + (h->sell_w > 0 ? 0 : ntl * 8) + 2 * h->n_rows * sv;
        return h->nnz + h->n_rows + ((h->n_rows + 255) / 256) * 8 + 2 * h->n_rows * sv;  // code + rowlen + bounds
    }
    return hipk_csr_spmv_bytes(h);
}

// ------------------------------------------------------------------ SpMV launch
// CAP (LDS product slots per tile): 1280 = 256 rows x 5 nnz, the 5-point stencil's tile, and it keeps the
// workgroup at 11 KB LDS => 8 workgroups per CU.  Denser tiles take the kernel's general path.
int hipk_launch_spmv(const hipk_csr_s *h, const hipk_spmv_args &a_, hipStream_t stream, hipk_spmv_profiler *prof) {
    hipk_spmv_args a = a_;
    const int ntiles = (int)((a.n + 255) / 256);
    const int grid = ((ntiles + 7) >> 3) << 3;
    a.tpart0 = h->tile_part;
    a.tpart1 = h->tile_part + 4 * (size_t)ntiles;
    if (h->op_cb != nullptr) {   // matrix-free operator (hipk_op_create): the caller's product, then the epilogue + combine
        HIPK_NOTE_KERNEL("%s", "operator callback + hipk_op_epilogue_kernel");
        if (h->op_cb(h->op_user, a.x, a.y) != 0) {
            hipk_set_error("the operator callback of a matrix-free handle failed");
            return HIPK_ERR_ARG;
        }
        if (a.mode != 0) {
            if (h->dtype == HIPK_F64)
                hipk_launch_timed(prof, HIPK_K_SPMV, hipk_op_epilogue_kernel<double>, ntiles, HIPK_THREADS, 0, stream, a);
            else
                hipk_launch_timed(prof, HIPK_K_SPMV, hipk_op_epilogue_kernel<float>, ntiles, HIPK_THREADS, 0, stream, a);
            if (!a.skip_combine && (a.mode & (HIPK_SPMV_DOT_W | HIPK_SPMV_DOT_YY)))
                hipk_launch_timed(prof, HIPK_K_AUX, hipk_tile_combine_kernel, (a.g + 3) / 4, HIPK_THREADS, 0, stream,
                                  (a.mode & HIPK_SPMV_DOT_W) ? a.tpart0 : nullptr, (a.mode & HIPK_SPMV_DOT_YY) ? a.tpart1 : nullptr,
                                  a.part0, a.part1, ntiles, a.ch / 256, a.g, a.stop_it, a.it);
        }
        HIPK_CHECK_HIP(hipGetLastError());
        return HIPK_OK;
    }
    // long-row matrices (mean row length >= 48, or rows that do not fit the LDS product buffer): row per wavefront
    const bool rowwave = h->n_rows > 0 && h->nnz / h->n_rows >= 48;
    if (!rowwave && h->n_huge > 0) {
        // pre-pass: the few rows that exceed the LDS product buffer, one wavefront each, raw sums into y
        hipk_spmv_args ah = a;
        ah.row_list = h->huge_rows;
        ah.n_list = h->n_huge;
        const int hgrid = (h->n_huge + 3) / 4;
        if (h->dtype == HIPK_F64)
            hipk_spmv_rowwave_kernel<double><<<hgrid, HIPK_THREADS, 0, stream>>>(ah);
        else
            hipk_spmv_rowwave_kernel<float><<<hgrid, HIPK_THREADS, 0, stream>>>(ah);
    }
    if (rowwave) {
        HIPK_NOTE_KERNEL("hipk_spmv_rowwave_kernel<%s>", h->dtype == HIPK_F64 ? "double" : "float");
        const int rgrid = (int)((a.n + 3) / 4);
        if (h->dtype == HIPK_F64)
            hipk_launch_timed(prof, HIPK_K_SPMV, hipk_spmv_rowwave_kernel<double>, rgrid, HIPK_THREADS, 0, stream, a);
        else
            hipk_launch_timed(prof, HIPK_K_SPMV, hipk_spmv_rowwave_kernel<float>, rgrid, HIPK_THREADS, 0, stream, a);
        if (a.mode & (HIPK_SPMV_DOT_W | HIPK_SPMV_DOT_YY)) {
            if (h->dtype == HIPK_F64)
                hipk_rowdot_kernel<double><<<ntiles, HIPK_THREADS, 0, stream>>>(a);
            else
                hipk_rowdot_kernel<float><<<ntiles, HIPK_THREADS, 0, stream>>>(a);
            if (!a.skip_combine) hipk_launch_timed(prof, HIPK_K_AUX, hipk_tile_combine_kernel, (a.g + 3) / 4, HIPK_THREADS, 0, stream, (a.mode & HIPK_SPMV_DOT_W) ? a.tpart0 : nullptr, (a.mode & HIPK_SPMV_DOT_YY) ? a.tpart1 : nullptr, a.part0,
                a.part1, ntiles, a.ch / 256, a.g, a.stop_it, a.it);
        }
        HIPK_CHECK_HIP(hipGetLastError());
        return HIPK_OK;
    }
    if (h->n_codes > 0 && h->path_override != 1) {
        a.code = h->code;
        a.rowlen = h->rowlen;
        a.dict_off = h->dict_off;
        a.dict_val = h->dict_val;
        a.n_codes = h->n_codes;
        a.code_cap = (h->max_tile_nnz + 32 + 15) & ~15;
        const int R = 1;
        const int nsuper = (ntiles + R - 1) / R;
        const int cgrid = ((nsuper + 7) >> 3) << 3;
        const size_t sv = (h->dtype == HIPK_F64) ? 8 : 4;
        const size_t lds = HIPK_CODED_MAX * (sv + 4) + (size_t)R * 16 + (size_t)R * a.code_cap;
        a.tile_off = h->tile_off;
        a.sell_w = h->sell_w;
        const bool sell = h->coded_layout >= 2;
        a.sell_vals = h->sell_vals;
        a.tile_ucode = h->tile_ucode;
        a.tile_wcode = h->tile_wcode ? h->tile_wcode : h->tile_ucode;   // two-rows-per-lane kernel: uniform AND masked tiles
        a.row_mask = h->row_mask;
        if (sell) {
            // persistent form: as many workgroups as can be resident (8 per CU), a multiple of 8 for the XCD mapping
            // exact tile size for the common stencil widths, run-time size otherwise
            const int tpc = a.ch / 256;
            void (*kern)(hipk_spmv_args) = nullptr;
            int lgrid = 0;
            bool chunked = false, strided = false;
            static const bool no_plan_cache = getenv("HIPK_SPMV_NO_PLAN_CACHE") != nullptr;
            const hipk_spmv_plan *pl = nullptr;
            for (int i = 0; i < h->n_plans && !no_plan_cache; ++i)
                if (h->plans[i].mode == a.mode && h->plans[i].ch == a.ch && h->plans[i].g == a.g) pl = &h->plans[i];
            if (pl != nullptr) {
                kern = (void (*)(hipk_spmv_args))pl->kern;
                lgrid = pl->lgrid;
                chunked = pl->chunked;
                strided = pl->strided;
                a.group_tiles = pl->group_tiles;
                memcpy(g_spmv_kernel, pl->name, sizeof(g_spmv_kernel));
            } else {
#define HIPK_PICK_LOOP_U(T, C, V, U)                                                                                      \
        (h->sell_w == 5 ? hipk_spmv_sell_loop_kernel<T, 5, C, V, U> : h->sell_w == 8 ? hipk_spmv_sell_loop_kernel<T, 8, C, V, U> \
         : h->sell_w == 4 ? hipk_spmv_sell_loop_kernel<T, 4, C, V, U> : hipk_spmv_sell_loop_kernel<T, 0, C, V, U>)
#define HIPK_PICK_LOOP_V(T, C, V) (h->tile_ucode ? HIPK_PICK_LOOP_U(T, C, V, true) : HIPK_PICK_LOOP_U(T, C, V, false))
#define HIPK_PICK_LOOP(T, C) (h->coded_layout == 3 ? HIPK_PICK_LOOP_V(T, C, true) : HIPK_PICK_LOOP_V(T, C, false))
                kern = (h->dtype == HIPK_F64) ? HIPK_PICK_LOOP(double, false) : HIPK_PICK_LOOP(float, false);
                const char *tname = h->dtype == HIPK_F64 ? "double" : "float";
                const int uw = (h->sell_w == 4 || h->sell_w == 5 || h->sell_w == 8) ? h->sell_w : 0;
                const char *uni = h->tile_ucode ? "true" : "false", *vls = h->coded_layout == 3 ? "true" : "false";
                HIPK_NOTE_KERNEL("hipk_spmv_sell_loop_kernel<%s,%d,false,%s,%s>", tname, uw, vls, uni);
                int occ = 0;  // resident workgroups per CU of this instantiation (register bound)
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, HIPK_THREADS, 0) != hipSuccess || occ < 1) occ = 4;
                const int slots = h->n_cu * occ;
                // one workgroup per reduction chunk when the chunks about fill the machine in one round
                // (default: the chunks fill at least half of the slots; a quarter where the two-rows-per-lane kernel applies --
                // N = 1.96 M Poisson: 28.2 -> 30.2 k CG it/s, no gain below a quarter)
                static const bool chunked_env = getenv("HIPK_SPMV_SELL_CHUNKED") != nullptr;
                const bool wide_ok = h->dtype == HIPK_F64 && h->tile_ucode && 2 * (h->n_uniform_tiles + h->n_masked_tiles) >= ntiles && h->coded_layout == 2 &&
                                     (h->sell_w == 4 || h->sell_w == 5 || h->sell_w == 8);
                const int cfac = (!chunked_env && wide_ok && h->sell_chunked == 2) ? 4 : h->sell_chunked;
                chunked = h->sell_chunked != 0 && tpc <= HIPK_SELL_MAX_TPC && a.g <= slots && cfac * a.g >= slots;
                static const bool no_mode = getenv("HIPK_SPMV_SELL_NO_MODE") != nullptr;
                const bool no_wide = getenv("HIPK_SPMV_SELL_NO_WIDE") != nullptr;  // read per launch: in-process A/B (tools/gmres_variants.py)
                constexpr int both = HIPK_SPMV_DOT_W | HIPK_SPMV_DOT_YY;
                // uniform tiles two rows per lane (hipk_spmv_sell_wide_kernel; fp64, most tiles uniform), mode bits compiled in for
                // the CG loop's form, the Arnoldi step's, BiCGStab's t = A s with <t, s> and <t, t> (TSL:925-927), plain y = A x
                auto pick_wide = [&](int st, char *pname, size_t cap) -> void (*)(hipk_spmv_args) {  // st = the kernel's WALK
                    void (*pk)(hipk_spmv_args) = nullptr;
#define HIPK_PICK_WIDE_S(M, S) \
        (h->sell_w == 5 ? hipk_spmv_sell_wide_kernel<5, M, S> : h->sell_w == 8 ? hipk_spmv_sell_wide_kernel<8, M, S> : hipk_spmv_sell_wide_kernel<4, M, S>)
#define HIPK_PICK_WIDE(M) (st == 1 ? HIPK_PICK_WIDE_S(M, 1) : HIPK_PICK_WIDE_S(M, 0))
                    pk = HIPK_PICK_WIDE(-1);
                    if (a.mode == HIPK_SPMV_DOT_W && !no_mode) pk = HIPK_PICK_WIDE(HIPK_SPMV_DOT_W);
                    if (a.mode == HIPK_SPMV_DOT_YY && !no_mode) pk = HIPK_PICK_WIDE(HIPK_SPMV_DOT_YY);
                    if (a.mode == both && !no_mode) pk = HIPK_PICK_WIDE(both);
                    if (a.mode == 0 && !no_mode) pk = HIPK_PICK_WIDE(0);
#undef HIPK_PICK_WIDE
#undef HIPK_PICK_WIDE_S
                    snprintf(pname, cap, "hipk_spmv_sell_wide_kernel<%d,%d,%d>", h->sell_w,  // the template arguments, as a profiler prints them
                             (a.mode >= 0 && a.mode <= both && !no_mode) ? a.mode : -1, st);
                    return pk;
                };
                // grouped walk of that kernel (WALK = 1: one workgroup per 4 consecutive tiles, tile sums through the combine kernel).
                // Taken (a) for a row block of FEW chunks of many tiles -- a rank of a row-partitioned system, which cannot fill the chip
                // with a workgroup per chunk and used to fall to the one-row-per-lane kernel (4 M rows with the chunk size of a
                // 4- / 8-rank weak-scaling run: 70.0 -> 65.4 / 72.5 -> 68.2 us per CG iteration) -- and (b) where a chunk holds 64 tiles
                // and more (N > 16 M on one device; per CG iteration 515 -> 483 us at N = 32 M, 1106 -> 980 us at N = 64 M, where a
                // chunk-walking workgroup fetched x three times) -- and, with four tiles per group, from 16 tiles per chunk: N = 8 M
                // 118.8 -> 117.1 us, N = 16 M equal, a rank's block of a 2-rank weak-scaling run (4 M rows, chunk 4096) 67.6 -> 66.5;
                // slower at N = 4 M (8 tiles per chunk, 56.6 -> 58.7: one more launch): not taken there.
                // HIPK_SPMV_SELL_STRIDED=0|1 forces (read per launch: in-process A/B, tools/walk_probe.py)
                if (wide_ok && !no_wide && h->sell_chunked != 0) {
                    const char *se = getenv("HIPK_SPMV_SELL_STRIDED");
                    if (se ? atoi(se) != 0 : tpc >= (chunked ? 16 : 32)) {
                        char pname[96];
                        kern = pick_wide(1, pname, sizeof(pname));
                        lgrid = hipk_xcd_grid((ntiles + HIPK_SELL_GROUP - 1) / HIPK_SELL_GROUP);
                        strided = true;
                        HIPK_NOTE_KERNEL("%s", pname);
                    }
                }
                // the same walk for the offset-coded form (value planes: variable-coefficient stencils), on the one-row-per-lane chunk
                // kernel with a grid of groups (hipk_spmv_args::group_tiles), from 32 tiles per chunk (N >= 16 M): CG per iteration
                // 336 -> 327 us at N = 16 M, 731 -> 704 at 32 M, 1573 -> 1418-1503 at 64 M (SpMV 873 -> 716 us).  NOT for pair codes
                // in fp32 storage, where the two-tiles-per-trip chunk kernel stays ahead (N = 64 M: 539 vs 562 us per iteration); the
                // switch forces it for any layout (tests, A/B)
                a.group_tiles = 0;
                if (!strided && h->sell_chunked != 0) {
                    const char *se = getenv("HIPK_SPMV_SELL_STRIDED");
                    if (se ? atoi(se) != 0 : (h->coded_layout == 3 && tpc >= 32)) {
                        kern = (h->dtype == HIPK_F64) ? HIPK_PICK_LOOP(double, true) : HIPK_PICK_LOOP(float, true);
                        a.group_tiles = HIPK_SELL_GROUP;
                        lgrid = hipk_xcd_grid((ntiles + HIPK_SELL_GROUP - 1) / HIPK_SELL_GROUP);
                        strided = true;
                        HIPK_NOTE_KERNEL("hipk_spmv_sell_loop_kernel<%s,%d,true,%s,%s>/groups", tname, uw, vls, uni);
                    }
                }
                if (strided) {
                    // kern, lgrid: set above
                } else if (chunked) {
                    kern = (h->dtype == HIPK_F64) ? HIPK_PICK_LOOP(double, true) : HIPK_PICK_LOOP(float, true);
                    lgrid = hipk_xcd_grid(a.g);
                    HIPK_NOTE_KERNEL("hipk_spmv_sell_loop_kernel<%s,%d,true,%s,%s>", tname, uw, vls, uni);
                    char pname[96];
                    // pair codes with an exact tile size: two tiles per loop trip (hipk_spmv_sell_pair_kernel)
                    static const bool no_pair = getenv("HIPK_SPMV_SELL_NO_PAIR") != nullptr;
                    if (!no_pair && h->coded_layout == 2 && (h->sell_w == 4 || h->sell_w == 5 || h->sell_w == 8)) {
#define HIPK_PICK_PAIR_U(T, U) \
        (h->sell_w == 5 ? hipk_spmv_sell_pair_kernel<T, 5, U> : h->sell_w == 8 ? hipk_spmv_sell_pair_kernel<T, 8, U> : hipk_spmv_sell_pair_kernel<T, 4, U>)
#define HIPK_PICK_PAIR(T) (h->tile_ucode ? HIPK_PICK_PAIR_U(T, true) : HIPK_PICK_PAIR_U(T, false))
                        void (*pk)(hipk_spmv_args) = (h->dtype == HIPK_F64) ? HIPK_PICK_PAIR(double) : HIPK_PICK_PAIR(float);
                        int pmode = -1;
                        // the CG loop's form (y = A x with <w, y>) of the 5-point fp64 stencil: mode bits compiled in
                        if (h->dtype == HIPK_F64 && h->sell_w == 5 && (a.mode == HIPK_SPMV_DOT_W || a.mode == HIPK_SPMV_DOT_YY) && !no_mode)
                            pmode = a.mode;
                        if (h->dtype == HIPK_F64 && h->sell_w == 5 && a.mode == HIPK_SPMV_DOT_W && !no_mode)
                            pk = h->tile_ucode ? hipk_spmv_sell_pair_kernel<double, 5, true, HIPK_SPMV_DOT_W>
                                               : hipk_spmv_sell_pair_kernel<double, 5, false, HIPK_SPMV_DOT_W>;
                        // the Arnoldi step's form (w = A v with ||w||^2, TSL:351-352)
                        if (h->dtype == HIPK_F64 && h->sell_w == 5 && a.mode == HIPK_SPMV_DOT_YY && !no_mode)
                            pk = h->tile_ucode ? hipk_spmv_sell_pair_kernel<double, 5, true, HIPK_SPMV_DOT_YY>
                                               : hipk_spmv_sell_pair_kernel<double, 5, false, HIPK_SPMV_DOT_YY>;
                        snprintf(pname, sizeof(pname), "hipk_spmv_sell_pair_kernel<%s,%d,%s,%d>", tname, h->sell_w, uni, pmode);
                        if (!no_wide && h->dtype == HIPK_F64 && h->tile_ucode && 2 * (h->n_uniform_tiles + h->n_masked_tiles) >= ntiles)
                            pk = pick_wide(0, pname, sizeof(pname));
                        int pocc = 0;  // the pair form holds more registers: take it only if the chunks still run as ONE round of workgroups
                        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&pocc, pk, HIPK_THREADS, 0) == hipSuccess &&
                            (pocc * h->n_cu >= a.g || pocc >= occ)) {
                            kern = pk;
                            HIPK_NOTE_KERNEL("%s", pname);
                        }
#undef HIPK_PICK_PAIR
#undef HIPK_PICK_PAIR_U
                    }
                } else {
                    lgrid = slots * h->sell_loop;
                    if (lgrid > ((ntiles + 7) >> 3) << 3) lgrid = ((ntiles + 7) >> 3) << 3;
                    lgrid = ((lgrid + 7) >> 3) << 3;
                }
#undef HIPK_PICK_LOOP
#undef HIPK_PICK_LOOP_V
#undef HIPK_PICK_LOOP_U
                if (!no_plan_cache && h->n_plans < (int)(sizeof(h->plans) / sizeof(h->plans[0]))) {
                    hipk_spmv_plan &np = h->plans[h->n_plans++];
                    np.mode = a.mode;
                    np.ch = a.ch;
                    np.g = a.g;
                    np.kern = (void *)kern;
                    np.lgrid = lgrid;
                    np.group_tiles = a.group_tiles;
                    np.chunked = chunked;
                    np.strided = strided;
                    memcpy(np.name, g_spmv_kernel, sizeof(np.name));
                }
            }
            hipk_launch_timed(prof, HIPK_K_SPMV, kern, lgrid, HIPK_THREADS, 0, stream, a);
            if ((!chunked || strided) && !a.skip_combine && (a.mode & (HIPK_SPMV_DOT_W | HIPK_SPMV_DOT_YY))) {
                hipk_launch_timed(prof, HIPK_K_AUX, hipk_tile_combine_kernel, (a.g + 3) / 4, HIPK_THREADS, 0, stream, (a.mode & HIPK_SPMV_DOT_W) ? a.tpart0 : nullptr, (a.mode & HIPK_SPMV_DOT_YY) ? a.tpart1 : nullptr,
                    a.part0, a.part1, ntiles, a.ch / 256, a.g, a.stop_it, a.it);
            }
            HIPK_CHECK_HIP(hipGetLastError());
            return HIPK_OK;
        }
        HIPK_NOTE_KERNEL("hipk_spmv_coded_kernel<%s,1>", h->dtype == HIPK_F64 ? "double" : "float");
        if (h->dtype == HIPK_F64)
            hipk_launch_timed(prof, HIPK_K_SPMV, hipk_spmv_coded_kernel<double, 1>, cgrid, HIPK_THREADS, lds, stream, a);
        else
            hipk_launch_timed(prof, HIPK_K_SPMV, hipk_spmv_coded_kernel<float, 1>, cgrid, HIPK_THREADS, lds, stream, a);
        if (!a.skip_combine && (a.mode & (HIPK_SPMV_DOT_W | HIPK_SPMV_DOT_YY))) {
            hipk_launch_timed(prof, HIPK_K_AUX, hipk_tile_combine_kernel, (a.g + 3) / 4, HIPK_THREADS, 0, stream, (a.mode & HIPK_SPMV_DOT_W) ? a.tpart0 : nullptr, (a.mode & HIPK_SPMV_DOT_YY) ? a.tpart1 : nullptr, a.part0,
                a.part1, ntiles, a.ch / 256, a.g, a.stop_it, a.it);
        }
        HIPK_CHECK_HIP(hipGetLastError());
        return HIPK_OK;
    }
    {
        const bool f64 = h->dtype == HIPK_F64, shortrows = h->max_row_len <= HIPK_LONG_ROW;
        const int cap = (f64 && h->max_tile_nnz <= 1280 && shortrows) ? 1280 : (h->max_tile_nnz <= 2048 && shortrows) ? 2048 : f64 ? 1280 : 2048;
        HIPK_NOTE_KERNEL("hipk_spmv_kernel<%s,%d,%s>", f64 ? "double" : "float", cap,
                         (h->max_tile_nnz <= cap && shortrows) ? "true" : "false");
    }
    if (h->dtype == HIPK_F64) {
        if (h->max_tile_nnz <= 1280 && h->max_row_len <= HIPK_LONG_ROW)
            hipk_launch_timed(prof, HIPK_K_SPMV, hipk_spmv_kernel<double, 1280, true>, grid, HIPK_THREADS, 0, stream, a);
        else if (h->max_tile_nnz <= 2048 && h->max_row_len <= HIPK_LONG_ROW)  // e.g. 7-point 3-D stencils (1792 per tile)
            hipk_launch_timed(prof, HIPK_K_SPMV, hipk_spmv_kernel<double, 2048, true>, grid, HIPK_THREADS, 0, stream, a);
        else
            hipk_launch_timed(prof, HIPK_K_SPMV, hipk_spmv_kernel<double, 1280, false>, grid, HIPK_THREADS, 0, stream, a);
    } else {
        if (h->max_tile_nnz <= 2048 && h->max_row_len <= HIPK_LONG_ROW)
            hipk_launch_timed(prof, HIPK_K_SPMV, hipk_spmv_kernel<float, 2048, true>, grid, HIPK_THREADS, 0, stream, a);
        else
            hipk_launch_timed(prof, HIPK_K_SPMV, hipk_spmv_kernel<float, 2048, false>, grid, HIPK_THREADS, 0, stream, a);
    }
    if (!a.skip_combine && (a.mode & (HIPK_SPMV_DOT_W | HIPK_SPMV_DOT_YY))) {
        hipk_launch_timed(prof, HIPK_K_AUX, hipk_tile_combine_kernel, (a.g + 3) / 4, HIPK_THREADS, 0, stream, (a.mode & HIPK_SPMV_DOT_W) ? a.tpart0 : nullptr, (a.mode & HIPK_SPMV_DOT_YY) ? a.tpart1 : nullptr, a.part0,
            a.part1, ntiles, a.ch / 256, a.g, a.stop_it, a.it);
    }
    HIPK_CHECK_HIP(hipGetLastError());
    return HIPK_OK;
}

static void hipk_fill_spmv_args(const hipk_csr_s *h, hipk_spmv_args &a, const void *x, void *y) {
    memset(&a, 0, sizeof(a));
    a.crow = h->crow;
    a.col = h->col;
    a.val = h->val;
    a.x = x;
    a.y = y;
    a.n = h->n_rows;
    a.ch = h->geom.ch;
    a.g = h->geom.g;
}

extern "C" int hipk_spmv(hipk_csr_t h, const void *x, void *y, hipk_stream_t stream) {
    HIPK_REQUIRE(h && x && y, HIPK_ERR_ARG, "null argument");
    HIPK_REQUIRE(hipk_aligned16(x) && hipk_aligned16(y), HIPK_ERR_ALIGN, "x/y must be 16-byte aligned");
    HIPK_REQUIRE(x != y, HIPK_ERR_ARG, "x and y must not alias");
    if (h->n_rows == 0) return HIPK_OK;
    hipk_spmv_args a;
    hipk_fill_spmv_args(h, a, x, y);
    return hipk_launch_spmv(h, a, (hipStream_t)stream);
}

extern "C" int hipk_spmv_dot(hipk_csr_t h, const void *x, void *y, const void *w, double *out_dev,
                             void *scratch_dev, hipk_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    HIPK_REQUIRE(h && x && y && w && out_dev && scratch_dev, HIPK_ERR_ARG, "null argument");
    HIPK_REQUIRE(hipk_aligned16(x) && hipk_aligned16(y) && hipk_aligned16(w), HIPK_ERR_ALIGN,
                 "x/y/w must be 16-byte aligned");
    HIPK_REQUIRE(x != y && w != y, HIPK_ERR_ARG, "y must not alias x or w");
    hipk_spmv_args a;
    hipk_fill_spmv_args(h, a, x, y);
    a.mode = HIPK_SPMV_DOT_W;
    a.w = w;
    a.part0 = (double *)scratch_dev;
    a.part1 = (double *)scratch_dev + HIPK_MAX_PARTS;
    if (h->n_rows > 0) {
        int rc = hipk_launch_spmv(h, a, stream);
        if (rc != HIPK_OK) return rc;
    }
    return hipk_launch_finish1(a.part0, h->n_rows > 0 ? a.g : 0, out_dev, stream);
}

extern "C" int hipk_spmv_ex(hipk_csr_t h, const void *x, void *y, int mode, const void *w, const void *bsub,
                            double *part0, double *part1, const int64_t *stop_dev, int64_t it,
                            hipk_stream_t stream) {
    HIPK_REQUIRE(h && x && y, HIPK_ERR_ARG, "null argument");
    HIPK_REQUIRE((mode & ~7) == 0, HIPK_ERR_ARG, "unknown mode bits");
    HIPK_REQUIRE(!(mode & HIPK_SPMV_DOT_W) || (w && part0), HIPK_ERR_ARG, "mode 1 needs w and part0");
    HIPK_REQUIRE(!(mode & HIPK_SPMV_DOT_YY) || part1, HIPK_ERR_ARG, "mode 2 needs part1");
    HIPK_REQUIRE(!(mode & HIPK_SPMV_RESID) || bsub, HIPK_ERR_ARG, "mode 4 needs bsub");
    HIPK_REQUIRE(hipk_aligned16(x) && hipk_aligned16(y) && hipk_aligned16(w) && hipk_aligned16(bsub), HIPK_ERR_ALIGN,
                 "vectors must be 16-byte aligned");
    HIPK_REQUIRE(x != y && w != y && bsub != y, HIPK_ERR_ARG, "y must not alias an input");
    if (h->n_rows == 0) return HIPK_OK;
    hipk_spmv_args a;
    hipk_fill_spmv_args(h, a, x, y);
    a.mode = mode;
    a.w = w;
    a.bsub = bsub;
    a.part0 = part0;
    a.part1 = part1;
    a.stop_it = stop_dev;
    a.it = it;
    return hipk_launch_spmv(h, a, (hipStream_t)stream);
}

#ifdef HIPK_GM_STAMPS
// diagnostic twin only: the per-wavefront phase stamps of the last hipk_spmv_sell_wide_kernel launch (hipk_coded.h)
extern "C" int hipk_debug_wide_stamps(unsigned long long *out, size_t count) {
    const size_t have = sizeof(hipk_wide_stamps) / sizeof(unsigned long long);
    HIPK_CHECK_HIP(hipDeviceSynchronize());
    HIPK_CHECK_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(hipk_wide_stamps), sizeof(unsigned long long) * (count < have ? count : have)));
    return HIPK_OK;
}
extern "C" int hipk_debug_wide_stamps_clear(void) {
    HIPK_CHECK_HIP(hipDeviceSynchronize());
    void *p = nullptr;
    HIPK_CHECK_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(hipk_wide_stamps)));
    HIPK_CHECK_HIP(hipMemset(p, 0, sizeof(hipk_wide_stamps)));
    return HIPK_OK;
}
#endif
