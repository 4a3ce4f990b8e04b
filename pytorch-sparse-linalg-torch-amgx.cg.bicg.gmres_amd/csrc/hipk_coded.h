// hipk_coded.h -- "coded" SpMV forms: one byte per entry for matrices that repeat few (col - row, value) pairs or few
// column offsets.
//
// Finite-difference / finite-volume matrices on structured grids -- what the reference's own builders produce
// (utils/matrix_utils.py:193-257 Poisson, examples ldc_solver_common.py:90-135 pressure matrix) -- repeat a handful
// of (column offset, coefficient) pairs millions of times.  hipk_csr_create builds, on the device, a dictionary of
// the distinct pairs (hash table in global memory, a workgroup-local LDS memo in front of it); when there are few
// enough and no row has more than 32 entries the handle keeps, next to the plain CSR arrays, one of three layouts
// (hipk_csr_s::coded_layout):
//   2  sliced-ELL planes of pair codes (<= 255 pairs): per 256-row tile, entries 4g..4g+3 of all rows as a plane of
//      dwords, the remaining one or two entries as byte planes; persistent kernel hipk_spmv_sell_loop_kernel.
//      SpMV streams 1 B per entry instead of 12 B + row pointers: 84 MB instead of 320 MB per product at N = 4M.
//   3  the same planes holding OFFSET codes only, plus value planes (variable-coefficient stencils): 9 B per entry.
//   1  pair codes in CSR order + byte row lengths (ragged matrices whose planes would be mostly padding, or exactly
//      256 pairs): hipk_spmv_coded_kernel, codes staged through LDS, row starts from a scan of the lengths.
// The arithmetic is the plain kernels', bit for bit: the dictionary returns the very same value bits and x[row + off]
// is x[col]; row sums are formed from rounded products in CSR order (rows have <= 32 entries, the spec's short-row
// rule), so the oracle needs no counterpart and every parity test runs on all paths.
#pragma once
#include "hipk_spmv.h"

#define HIPK_CODED_MAX 256   // dictionary entries (uint8 codes)
#define HIPK_DICT_SLOTS 2048  // open-addressed build table (power of two, >= 8 x HIPK_CODED_MAX)

#ifdef __HIPCC__
// ------------------------------------------------------------------ dictionary construction (handle creation)
struct hipk_dict_table {
    unsigned long long key[HIPK_DICT_SLOTS];   // 0 = empty, else hash of the pair
    unsigned long long bits[HIPK_DICT_SLOTS];  // payload: value bits (fp32 zero-extended)
    int off[HIPK_DICT_SLOTS];                  // payload: col - row
    int slot_code[HIPK_DICT_SLOTS];            // filled by the host between the two passes
    int count;                                 // distinct pairs inserted
    int overflow;                              // more than HIPK_CODED_MAX pairs: give up
    int fail;                                  // encode pass: pair not found / hash collision
};

__device__ __forceinline__ unsigned long long hipk_pair_hash(int off, unsigned long long bits) {
    unsigned long long h = bits ^ ((unsigned long long)(unsigned int)off * 0x9E3779B97F4A7C15ull);
    h ^= h >> 31;
    h *= 0xBF58476D1CE4E5B9ull;
    h ^= h >> 29;
    h *= 0x94D049BB133111EBull;
    h ^= h >> 32;
    return h ? h : 1ull;
}

template <typename T>
__device__ __forceinline__ unsigned long long hipk_value_bits(T v);
template <>
__device__ __forceinline__ unsigned long long hipk_value_bits<double>(double v) {
    return (unsigned long long)__double_as_longlong(v);
}
template <>
__device__ __forceinline__ unsigned long long hipk_value_bits<float>(float v) {
    return (unsigned long long)__float_as_uint(v);
}

// Workgroup-local memo of the global table (LDS, open addressing on the same hash): 20 M entries asking the same
// five global slots would serialise on one L2 channel (3.4 ms per pass at N = 4 M).  An entry first looks its
// hash up here; only the first sighting per workgroup goes to the global table.  val: pass 1 stores 1 ("known to
// be in the global table"), pass 2 stores 1 + code.
#define HIPK_MEMO_SLOTS 1024
struct hipk_dict_memo {
    unsigned long long key[HIPK_MEMO_SLOTS];
    unsigned long long bits[HIPK_MEMO_SLOTS];  // payload, compared exactly: a colliding hash is a miss, never a wrong code
    int off[HIPK_MEMO_SLOTS];
    int val[HIPK_MEMO_SLOTS];
};
__device__ __forceinline__ void hipk_memo_clear(hipk_dict_memo &m) {
    for (int i = threadIdx.x; i < HIPK_MEMO_SLOTS; i += blockDim.x) {
        m.key[i] = 0ull;
        m.val[i] = 0;
    }
    __syncthreads();
}
// returns val (> 0) if the pair is memoised, else 0
__device__ __forceinline__ int hipk_memo_find(const hipk_dict_memo &m, unsigned long long h, int off,
                                              unsigned long long bits) {
    unsigned int s = (unsigned int)(h >> 20) & (HIPK_MEMO_SLOTS - 1);
    for (int probe = 0; probe < 16; ++probe) {
        const unsigned long long k = ((volatile const unsigned long long *)m.key)[s];
        if (k == h) {
            const int v = ((volatile const int *)m.val)[s];  // 0 while the owner is still publishing: a miss
            if (v != 0 && ((volatile const unsigned long long *)m.bits)[s] == bits && ((volatile const int *)m.off)[s] == off)
                return v;
            return 0;
        }
        if (k == 0) return 0;
        s = (s + 1) & (HIPK_MEMO_SLOTS - 1);
    }
    return 0;
}
__device__ __forceinline__ void hipk_memo_put(hipk_dict_memo &m, unsigned long long h, int off, unsigned long long bits,
                                              int val) {
    unsigned int s = (unsigned int)(h >> 20) & (HIPK_MEMO_SLOTS - 1);
    for (int probe = 0; probe < 16; ++probe) {
        const unsigned long long k = atomicCAS(&m.key[s], 0ull, h);
        if (k == 0) {  // this thread owns the slot: payload first, then the value that makes it visible
            ((volatile unsigned long long *)m.bits)[s] = bits;
            ((volatile int *)m.off)[s] = off;
            __threadfence_block();
            ((volatile int *)m.val)[s] = val;
            return;
        }
        if (k == h) return;  // someone else memoises this hash (same pair, or a collision that stays a miss)
        s = (s + 1) & (HIPK_MEMO_SLOTS - 1);
    }  // memo full around this hash: the entry simply keeps asking the global table
}

// pass 1: insert every distinct (col - row, value) pair.  A global slot is claimed with one CAS.
// OFFS_ONLY: the dictionary holds column offsets only (values stay per entry: variable-coefficient stencils)
template <typename T, bool OFFS_ONLY>
__global__ __launch_bounds__(HIPK_THREADS) void hipk_dict_insert_kernel(const int *__restrict__ crow,
                                                                        const int *__restrict__ col,
                                                                        const T *__restrict__ val, int64_t n_rows,
                                                                        hipk_dict_table *tb) {
    __shared__ hipk_dict_memo memo;
    hipk_memo_clear(memo);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    // the overflow flag is read once, with an ordinary (cacheable) load: an atomic load per row was 4 M round trips to
    // one address.  A stale "no overflow" only means this thread finishes its few rows.
    if (((volatile const int *)&tb->overflow)[0]) return;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_rows; r += stride) {
        const int lo = crow[r], hi = crow[r + 1];
        for (int j = lo; j < hi; ++j) {
            const int off = col[j] - (int)r;
            const unsigned long long bits = OFFS_ONLY ? 0ull : hipk_value_bits<T>(val[j]);
            const unsigned long long h = hipk_pair_hash(off, bits);
            if (hipk_memo_find(memo, h, off, bits)) continue;
            unsigned int s = (unsigned int)h & (HIPK_DICT_SLOTS - 1);
            for (int probe = 0; probe < HIPK_DICT_SLOTS; ++probe) {
                // plain (cacheable) look-up: a stale "empty" only costs a CAS, which returns the true owner
                unsigned long long k = ((volatile const unsigned long long *)tb->key)[s];
                if (k == 0) {
                    k = atomicCAS(&tb->key[s], 0ull, h);
                    if (k == 0) {  // claimed: publish the payload (read only by the NEXT kernel)
                        tb->bits[s] = bits;
                        tb->off[s] = off;
                        if (atomicAdd(&tb->count, 1) + 1 > HIPK_CODED_MAX) {
                            __hip_atomic_store(&tb->overflow, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            return;
                        }
                        break;
                    }
                }
                if (k == h) break;
                s = (s + 1) & (HIPK_DICT_SLOTS - 1);
            }
            hipk_memo_put(memo, h, off, bits, 1);
        }
    }
}

// global look-up of a pair: slot index, or -1.  The payload is compared exactly: a hash collision or a value that
// changed between the passes is reported as "not found".
template <typename T>
__device__ __forceinline__ int hipk_dict_lookup(const hipk_dict_table *tb, unsigned long long h, int off,
                                                unsigned long long bits) {
    unsigned int s = (unsigned int)h & (HIPK_DICT_SLOTS - 1);
    for (int probe = 0; probe < HIPK_DICT_SLOTS; ++probe) {
        const unsigned long long k = tb->key[s];
        if (k == h) return (tb->bits[s] == bits && tb->off[s] == off) ? (int)s : -1;
        if (k == 0) return -1;
        s = (s + 1) & (HIPK_DICT_SLOTS - 1);
    }
    return -1;
}

// pass 2: code[j] = dictionary index of entry j, rowlen[r] = entries of row r (CSR-ordered layout)
template <typename T>
__global__ __launch_bounds__(HIPK_THREADS) void hipk_dict_encode_kernel(const int *__restrict__ crow,
                                                                        const int *__restrict__ col,
                                                                        const T *__restrict__ val, int64_t n_rows,
                                                                        hipk_dict_table *tb,
                                                                        unsigned char *__restrict__ code,
                                                                        unsigned char *__restrict__ rowlen) {
    __shared__ hipk_dict_memo memo;
    hipk_memo_clear(memo);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_rows; r += stride) {
        const int lo = crow[r], hi = crow[r + 1];
        rowlen[r] = (unsigned char)(hi - lo);
        for (int j = lo; j < hi; ++j) {
            const int off = col[j] - (int)r;
            const unsigned long long bits = hipk_value_bits<T>(val[j]);
            const unsigned long long h = hipk_pair_hash(off, bits);
            int c1 = hipk_memo_find(memo, h, off, bits);  // 1 + code
            if (c1 == 0) {
                const int found = hipk_dict_lookup<T>(tb, h, off, bits);
                if (found < 0) {
                    tb->fail = 1;
                    code[j] = 0;
                    continue;
                }
                c1 = 1 + tb->slot_code[found];
                hipk_memo_put(memo, h, off, bits, c1);
            }
            code[j] = (unsigned char)(c1 - 1);
        }
    }
}

// ------------------------------------------------------------------ the SpMV kernel
// dynamic LDS layout: dval[256] T | doff[256] int | wtot[R][4] int | R code buffers of a.code_cap bytes
template <typename T, int R>
__global__ __launch_bounds__(HIPK_THREADS) void hipk_spmv_coded_kernel(hipk_spmv_args a) {
    const int ntiles = (int)((a.n + HIPK_TILE - 1) / HIPK_TILE);
    const int nsuper = (ntiles + R - 1) / R;
    const int st = hipk_xcd_tile(blockIdx.x, nsuper);
    if (st < 0) return;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T *dval = (T *)smem;
    int *doff = (int *)(smem + HIPK_CODED_MAX * sizeof(T));
    int *wtot = doff + HIPK_CODED_MAX;
    unsigned char *cbuf = (unsigned char *)(wtot + R * 4);

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    const int *__restrict__ crow = a.crow;
    const unsigned char *__restrict__ code = a.code;
    const unsigned char *__restrict__ rowlen = a.rowlen;
    const T *__restrict__ x = (const T *)a.x;
    T *__restrict__ y = (T *)a.y;
    const int mode = a.mode;

    int64_t r0[R];
    int nr[R], len[R], head[R];  // head: offset of the tile's first code inside its LDS buffer
    T wrow[R], brow[R], drow[R];
    uint4 cv[R];
    int nvec[R];
    const unsigned char *cbase[R];
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const int tile = st * R + i;
        r0[i] = (int64_t)tile * HIPK_TILE;
        nr[i] = tile < ntiles ? (int)((a.n - r0[i] < HIPK_TILE) ? (a.n - r0[i]) : HIPK_TILE) : 0;
        len[i] = 0;
        wrow[i] = (T)0;
        brow[i] = (T)0;
        drow[i] = (T)0;
        nvec[i] = 0;
        head[i] = 0;
        cbase[i] = code;
        if (nr[i] > 0) {
            // tile bounds: uniform addresses -> scalar loads
            const int j0 = __builtin_amdgcn_readfirstlane(crow[r0[i]]);
            const int j1 = __builtin_amdgcn_readfirstlane(crow[r0[i] + nr[i]]);
            const int a0 = j0 & ~15;
            head[i] = j0 - a0;
            nvec[i] = (j1 - a0 + 15) >> 4;
            cbase[i] = code + a0;
            if (t < nr[i]) {
                len[i] = rowlen[r0[i] + t];
                if (mode & HIPK_SPMV_DOT_W) wrow[i] = ((const T *)a.w)[r0[i] + t];
                if (mode & HIPK_SPMV_RESID) brow[i] = ((const T *)a.bsub)[r0[i] + t];
                if (mode & HIPK_SPMV_SCALE) drow[i] = ((const T *)a.dscale)[r0[i] + t];
            }
            if (t < nvec[i]) cv[i] = ((const uint4 *)cbase[i])[t];
        }
    }
    // the stop word is read only now: the tile's loads above are already in flight (they touch valid memory
    // whatever the answer is), nothing has been stored yet
    if (a.stop_it != nullptr && a.it >= *a.stop_it) return;
    if (t < a.n_codes) {
        dval[t] = ((const T *)a.dict_val)[t];
        doff[t] = a.dict_off[t];
    }
    int excl[R];
#pragma unroll
    for (int i = 0; i < R; ++i) {
        unsigned char *cb = cbuf + (size_t)i * a.code_cap;
        if (t < nvec[i]) ((uint4 *)cb)[t] = cv[i];
        for (int v = t + HIPK_THREADS; v < nvec[i]; v += HIPK_THREADS) ((uint4 *)cb)[v] = ((const uint4 *)cbase[i])[v];
        int s = len[i];  // inclusive scan of the row lengths over the wavefront
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int u = __shfl_up(s, o);
            if (lane >= o) s += u;
        }
        excl[i] = s - len[i];
        if (lane == 63) wtot[i * 4 + wave] = s;
    }
    __syncthreads();

#pragma unroll
    for (int i = 0; i < R; ++i) {
        if (nr[i] == 0) continue;
        const unsigned char *cb = cbuf + (size_t)i * a.code_cap;
        int lo = head[i] + excl[i];
        for (int w = 0; w < wave; ++w) lo += wtot[i * 4 + w];
        const int64_t row = r0[i] + t;
        T s = (T)0;
        const int n_ent = len[i];
        for (int k0 = 0; k0 < n_ent; k0 += 8) {
            T xv[8];
            int cc[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (k0 + k < n_ent) {
                    cc[k] = cb[lo + k0 + k];
                    xv[k] = x[row + doff[cc[k]]];
                }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (k0 + k < n_ent) {
                    const T p = dval[cc[k]] * xv[k];
                    s = s + p;
                }
            }
        }
        double d0 = 0.0, d1 = 0.0;
        if (t < nr[i]) {
            T out = s;
            if (mode & HIPK_SPMV_RESID) out = brow[i] - out;
            if (mode & HIPK_SPMV_SCALE) out = drow[i] * out;
            y[row] = out;
            if (mode & HIPK_SPMV_DOT_W) d0 = (double)wrow[i] * (double)out;
            if (mode & HIPK_SPMV_DOT_YY) d1 = (double)out * (double)out;
        }
        const size_t tp = (size_t)(st * R + i) * 4 + wave;
        if (mode & HIPK_SPMV_DOT_W) {
d0 = hipk_wave_sum(d0);
            if (lane == 0) a.tpart0[tp] = d0;
        }
        if (mode & HIPK_SPMV_DOT_YY) {
d1 = hipk_wave_sum(d1);
            if (lane == 0) a.tpart1[tp] = d1;
        }
    }
}

// ================================================================== sliced-ELL layout of the codes
// The CSR-ordered code bytes above still make a workgroup wait three times in a row (tile bounds -> code bytes ->
// x gather) and cost a scan, an LDS copy and a barrier per tile.  The sliced-ELL layout stores the codes of a
// 256-row tile column-wise: with W = the longest row of the tile, entry k of row t lives in
//     dword plane k/4 (1 KB: one little-endian dword per row, byte k%4)           for k < 4 D
//     byte plane k - 4 D (256 B: one byte per row), after the D dword planes       for the remaining Bp <= 2
// where D = W/4 and Bp = W%4, except that W%4 == 3 takes one more dword plane instead of three byte planes.
// A 5-point stencil tile is one dword plane + one byte plane = 1280 B: two coalesced loads per thread bring
// a row's five codes.  Rows shorter than W are padded with HIPK_SELL_PAD, which is skipped by a select, never
// multiplied, so row sums are formed from exactly the CSR entries in CSR order.  A tile's size is U = 4 D + Bp
// units of 256 B (tile_off = prefix sum of U); the layout is built only when it takes at most 2 x nnz bytes.
#define HIPK_SELL_PAD 255  // dictionary limited to 255 entries in this layout

static inline int hipk_sell_units(int w) {  // host: units of 256 B of a tile whose longest row has w entries
    const int d = w / 4 + ((w % 4 == 3) ? 1 : 0);
    const int bp = (w % 4 == 3) ? 0 : w % 4;
    return 4 * d + bp;
}

__global__ __launch_bounds__(HIPK_THREADS) void hipk_tile_width_kernel(const int *__restrict__ crow, int64_t n_rows,
                                                                       int *__restrict__ tile_w) {
    __shared__ int wmax[HIPK_THREADS / 64];
    const int64_t r = (int64_t)blockIdx.x * HIPK_TILE + threadIdx.x;
    int len = (r < n_rows) ? crow[r + 1] - crow[r] : 0;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const int u = __shfl_down(len, o);
        len = u > len ? u : len;
    }
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = len;
    __syncthreads();
    if (threadIdx.x == 0) {
        int m = wmax[0];
        for (int w = 1; w < HIPK_THREADS / 64; ++w) m = wmax[w] > m ? wmax[w] : m;
        tile_w[blockIdx.x] = m;
    }
}

// Uniform tiles: on a constant-coefficient stencil almost every tile's 256 rows carry the SAME code bytes (all but
// the tiles that contain a grid-line end).  ucode[tile] = those bytes (two groups of four, padding 0xFF) when the
// tile has at most two groups and all rows agree, else 0 (no valid tile packs to 0: codes of a row are distinct).
// The SpMV then takes a uniform tile's codes from one scalar load and never reads its planes.
__global__ __launch_bounds__(HIPK_THREADS) void hipk_tile_uniform_kernel(const unsigned char *__restrict__ code,
                                                                         const int *__restrict__ tile_off, int ntiles,
                                                                         unsigned long long *__restrict__ ucode,
                                                                         int *__restrict__ count) {
    const int t = threadIdx.x;
    __shared__ unsigned first[2];
    int n_ok = 0, units_ok = 0;  // thread 0: this workgroup's tally (one pair of atomics per 16 tiles, not per tile)
    for (int tl = blockIdx.x * 16; tl < ntiles && tl < blockIdx.x * 16 + 16; ++tl) {
    const int o0 = tile_off[tl], o1 = tile_off[tl + 1];
    const int D = (o1 - o0) >> 2, Bp = (o1 - o0) & 3;
    const unsigned char *tp = code + (size_t)o0 * HIPK_TILE;
    unsigned c[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        unsigned w = 0xFFFFFFFFu;
        if (g < D) {
            w = ((const unsigned *)tp)[g * HIPK_TILE + t];
        } else if (g == D) {
            const unsigned char *bp = tp + (size_t)D * 1024 + t;
            if (Bp >= 1) w = (w & 0xFFFFFF00u) | bp[0];
            if (Bp >= 2) w = (w & 0xFFFF00FFu) | ((unsigned)bp[HIPK_TILE] << 8);
        }
        c[g] = w;
    }
    if (t == 0) {
        first[0] = c[0];
        first[1] = c[1];
    }
    __syncthreads();
    const int groups = D + (Bp > 0 ? 1 : 0);
    const int same = __syncthreads_and(c[0] == first[0] && c[1] == first[1]);  // also fences `first` for the next tile
    if (t == 0) {
        const bool ok = same && groups <= 2 && groups >= 1;
        ucode[tl] = ok ? ((unsigned long long)c[0] | ((unsigned long long)c[1] << 32)) : 0ull;
        if (ok) {
            ++n_ok;
            units_ok += o1 - o0;
        }
    }
    }
    if (t == 0 && n_ok > 0) {
        atomicAdd(count, n_ok);
        atomicAdd(count + 1, units_ok);
    }
}

// Masked tiles (round 3).  The stamps of the two-rows-per-lane kernel (profiles/r03_spmv_wide_stamps.md) show the tiles that
// contain a grid-line end -- one in eight on the 2000-wide Poisson grid, taken one row per lane by the whole workgroup -- as what
// spreads the workgroups' run times (median 10.5 us, last 15.2 us).  Their rows do not carry the same codes, but they differ only
// by WHICH entries of one pattern they have (the row at a line's end lacks the east neighbour, the next one the west): each row's
// code list is a SUBSEQUENCE of the tile's longest row's.  wcode[tile] = that pattern with byte 7 = HIPK_SELL_MASKED, mask[row] =
// which of its entries the row has; the kernel forms all products of the pattern and SKIPS the absent ones with a select (same
// products, same order of additions as the one-row-per-lane path: same bits).  Conditions: full tile, <= 7 entries per row,
// every load of the pattern in range for every row pair (so not the tiles at the first / last grid line of the matrix).
#define HIPK_SELL_MASKED 254u
__global__ __launch_bounds__(HIPK_THREADS) void hipk_tile_masked_kernel(const unsigned char *__restrict__ code,
                                                                        const int *__restrict__ tile_off, int ntiles, int64_t n_rows,
                                                                        const int *__restrict__ dict_off,
                                                                        const unsigned long long *__restrict__ ucode,
                                                                        unsigned long long *__restrict__ wcode,
                                                                        unsigned char *__restrict__ mask, int *__restrict__ count) {
    const int t = threadIdx.x;
    __shared__ unsigned best[2];   // packed (length << 8 | 255 - t) of the longest row, its two code words below
    __shared__ unsigned ucw[2];
    int n_ok = 0, units_ok = 0;
    for (int tl = blockIdx.x * 16; tl < ntiles && tl < blockIdx.x * 16 + 16; ++tl) {
        const unsigned long long uc = ucode[tl];
        if (uc != 0ull) {   // uniform: the same word, every entry present
            if (t == 0) wcode[tl] = uc;
            mask[(size_t)tl * HIPK_TILE + t] = 0xFF;
            continue;
        }
        const int o0 = tile_off[tl], o1 = tile_off[tl + 1];
        const int D = (o1 - o0) >> 2, Bp = (o1 - o0) & 3;
        const unsigned char *tp = code + (size_t)o0 * HIPK_TILE;
        unsigned c[2];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            unsigned w = 0xFFFFFFFFu;
            if (g < D) {
                w = ((const unsigned *)tp)[g * HIPK_TILE + t];
            } else if (g == D) {
                const unsigned char *bp = tp + (size_t)D * 1024 + t;
                if (Bp >= 1) w = (w & 0xFFFFFF00u) | bp[0];
                if (Bp >= 2) w = (w & 0xFFFF00FFu) | ((unsigned)bp[HIPK_TILE] << 8);
            }
            c[g] = w;
        }
        const int groups = D + (Bp > 0 ? 1 : 0);
        int len = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (((c[k >> 2] >> ((k & 3) * 8)) & 0xFFu) != HIPK_SELL_PAD) len = k + 1;   // codes of a row are packed from entry 0
        if (t == 0) best[0] = 0u;
        __syncthreads();
        atomicMax(&best[0], ((unsigned)len << 8) | (unsigned)(255 - t));
        __syncthreads();
        const int who = 255 - (int)(best[0] & 0xFFu), ulen = (int)(best[0] >> 8);
        if (t == who) {
            ucw[0] = c[0];
            ucw[1] = c[1];
        }
        __syncthreads();
        // greedy subsequence match of this row's codes against the pattern
        unsigned m = 0u;
        int j = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const unsigned uk = (ucw[k >> 2] >> ((k & 3) * 8)) & 0xFFu;
            const unsigned cj = j < 8 ? ((c[j >> 2] >> ((j & 3) * 8)) & 0xFFu) : HIPK_SELL_PAD;
            if (k < ulen && j < len && cj == uk) {
                m |= 1u << k;
                ++j;
            }
        }
        int ok = (j == len) && groups >= 1 && groups <= 2 && ulen <= 7 && (int64_t)(tl + 1) * HIPK_TILE <= n_rows;
        // every 16-byte load of the pattern in range for every row pair of the tile
        if (t < ulen) {
            const unsigned uk = (ucw[t >> 2] >> ((t & 3) * 8)) & 0xFFu;
            const int64_t off = dict_off[uk];
            if ((int64_t)tl * HIPK_TILE + off < 0 || (int64_t)tl * HIPK_TILE + HIPK_TILE - 1 + off > n_rows - 1) ok = 0;
        }
        const int all = __syncthreads_and(ok);
        mask[(size_t)tl * HIPK_TILE + t] = all ? (unsigned char)m : 0;
        if (t == 0) {
            wcode[tl] = all ? (((unsigned long long)ucw[0] | ((unsigned long long)ucw[1] << 32)) & 0x00FFFFFFFFFFFFFFull) |
                                  ((unsigned long long)HIPK_SELL_MASKED << 56)
                            : 0ull;
            if (all) {
                ++n_ok;
                units_ok += o1 - o0;
            }
        }
    }
    if (t == 0 && n_ok > 0) {
        atomicAdd(count, n_ok);
        atomicAdd(count + 1, units_ok);
    }
}

// code of row r, entry k -> its byte in r's tile (the planes are prefilled with HIPK_SELL_PAD)
template <typename T, bool OFFS_ONLY>
__global__ __launch_bounds__(HIPK_THREADS) void hipk_dict_encode_sell_kernel(
    const int *__restrict__ crow, const int *__restrict__ col, const T *__restrict__ val, int64_t n_rows,
    hipk_dict_table *tb, const int *__restrict__ tile_off, unsigned char *__restrict__ code, T *__restrict__ vals) {
    __shared__ hipk_dict_memo memo;
    hipk_memo_clear(memo);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_rows; r += stride) {
        const int lo = crow[r], hi = crow[r + 1];
        const int o0 = tile_off[r >> 8], units = tile_off[(r >> 8) + 1] - o0;
        const int D = units >> 2;
        unsigned char *tilep = code + (size_t)o0 * HIPK_TILE;
        const int t = (int)(r & 255);
        for (int j = lo; j < hi; ++j) {
            const int off = col[j] - (int)r;
            const unsigned long long bits = OFFS_ONLY ? 0ull : hipk_value_bits<T>(val[j]);
            const unsigned long long h = hipk_pair_hash(off, bits);
            const int k = j - lo;
            int c1 = hipk_memo_find(memo, h, off, bits);  // 1 + code
            if (c1 == 0) {
                const int found = hipk_dict_lookup<T>(tb, h, off, bits);
                if (found >= 0) {
                    c1 = 1 + tb->slot_code[found];
                    hipk_memo_put(memo, h, off, bits, c1);
                }
            }
            if (c1 == 0 || k >= 4 * D + (units & 3)) {
                tb->fail = 1;
            } else {
                const size_t pos = (k < 4 * D) ? (size_t)(k >> 2) * 1024 + (size_t)t * 4 + (k & 3)
                                               : (size_t)D * 1024 + (size_t)(k - 4 * D) * HIPK_TILE + t;
                tilep[pos] = (unsigned char)(c1 - 1);
                // value planes (offset-coded layout): plane k of the tile, one value per row; a tile of U units holds
                // exactly U entries per row, so its planes start at element tile_off * 256
                if (OFFS_ONLY) vals[(size_t)o0 * HIPK_TILE + (size_t)k * HIPK_TILE + t] = val[j];
            }
        }
    }
}

// Persistent kernel.  A one-tile-per-workgroup kernel is bound by what every workgroup does ONCE (kernel
// arguments, dictionary -> LDS, barrier, stop word) plus two dependent waits per tile (codes, x gather): ~3 us
// per wavefront for ~5 KB of traffic.  Here the grid is the number of resident workgroups; each walks its share
// of the tiles of its XCD's eighth of the matrix with the NEXT tile's codes (two packed dwords = 8 entries) and
// epilogue operands already requested while the current tile's x gathers are in flight.  The loop body is branch
// free: padding codes read dictionary slot 255 (offset 0, value 0) and are kept out of the sum by a select.
// UNITS > 0 instantiates the exact tile size of a uniform matrix (5: 5-point stencil; 4: width 3-4; 8: width 7-8),
// UNITS == 0 reads tile_off and walks further groups of four codes at run time.
// Requires n_rows <= n_cols (padding lanes read x[min(row, n_rows-1)]) and n_cols * sizeof(T) < 4 GiB (32-bit byte
// offsets: one shift + one add per gather).
//
// CHUNKED = false: workgroup b walks tiles idx, idx + gp, .. of its XCD's eighth and stores the four wavefront sums
//   of each tile's fused dot; hipk_tile_combine_kernel folds them into chunk partials afterwards.
// CHUNKED = true (large systems: about as many reduction chunks as resident workgroups): one workgroup per
//   reduction chunk, its tiles in ascending order; the wavefront sums stay in LDS and the workgroup itself forms
//   the chunk partial with the spec's fold (hipk_wave_fold) -- no combine launch (4.9 us per CG iteration).
#ifndef HIPK_SELL_GROUP
// tiles per workgroup of the grouped walk.  Same box, library twins (tools/r02_call46.sh), CG per iteration at N = 32 M / 64 M and on a
// rank-shaped block of 4 M rows: 2 tiles 496 / 1030 / 72.2 us, 4 tiles 481 / 998 / 69.2, 8 tiles 489 / 1007 / 69.2, 16 tiles 503 / 1022 / 72.0
#define HIPK_SELL_GROUP 4
#endif
#define HIPK_SELL_MAX_TPC 128  // tiles per chunk the chunked form holds in LDS (chunks up to 32768 rows: N = 64 M)
// VALS = true: offset-coded layout -- the dictionary holds column offsets only, the values come from per-tile value
//   planes (plane k = the k-th entry of every row, coalesced, non-temporal): 9 instead of 12 bytes per entry and no
//   row pointers, for stencils with variable coefficients.
template <typename T, int UNITS, bool CHUNKED, bool VALS, bool UNI = false>
__global__ __launch_bounds__(HIPK_THREADS) HIPK_SGPR80 void hipk_spmv_sell_loop_kernel(hipk_spmv_args a) {
    constexpr int G0 = UNITS == 0 ? 2 : (UNITS + 3) / 4;  // groups of four codes held in registers (<= 2)
    static_assert(UNITS == 0 || UNITS <= 8, "exact instantiations cover up to 8 entries per row");
    const int ntiles = (int)((a.n + HIPK_TILE - 1) / HIPK_TILE);
    const int per = (ntiles + 7) >> 3;       // tiles per XCD eighth
    const int gp = (int)gridDim.x >> 3;      // workgroups per XCD (grid is a multiple of 8)
    const int xcd = blockIdx.x & 7;
    int idx = blockIdx.x >> 3;               // position inside the eighth; advances by gp
    const int tpc = a.ch / HIPK_TILE;        // tiles per reduction chunk
    // CHUNKED with a.group_tiles > 0 (a scalar: no further instantiations): the workgroup takes a GROUP of that many consecutive
    // tiles instead of a reduction chunk, on a grid of groups, and leaves the fold to the combine kernel -- the grouped walk of
    // hipk_spmv_sell_wide_kernel (see there: chunks of many tiles re-fetch x[row +- nx] from beyond L2); the launcher takes it
    // for the offset-coded form (value planes) from N = 16 M, and for any layout when forced (tests)
    const int gt = CHUNKED ? a.group_tiles : 0;
    const int wtiles = gt > 0 ? gt : tpc;  // tiles per workgroup
    const int chunk = CHUNKED ? hipk_xcd_chunk(blockIdx.x, gt > 0 ? (ntiles + gt - 1) / gt : a.g) : 0;
    if (CHUNKED && chunk < 0) return;
    const int t_first = chunk * wtiles;
    const int t_end = (t_first + wtiles < ntiles) ? t_first + wtiles : ntiles;  // CHUNKED: this workgroup's tiles
    __shared__ double wsum0[CHUNKED ? HIPK_SELL_MAX_TPC * 4 : 1];
    __shared__ double wsum1[CHUNKED ? HIPK_SELL_MAX_TPC * 4 : 1];

    __shared__ T dval[HIPK_CODED_MAX];
    __shared__ int doff[HIPK_CODED_MAX];
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    const unsigned char *__restrict__ code = a.code;
    const char *__restrict__ xb = (const char *)a.x;
    T *__restrict__ y = (T *)a.y;
    const int mode = a.mode;
    const int n32 = (int)a.n;

    T dv = (T)0;
    int dofs = 0;
    if (t < a.n_codes) {
        dv = ((const T *)a.dict_val)[t];
        dofs = a.dict_off[t];
    }
    // group g of a tile with D dword planes and Bp byte planes: four codes packed in a dword, padding = 0xFF
    auto load_group = [&](const unsigned char *tp, int D, int Bp, int g) -> unsigned {
        if (g < D) return ((const unsigned *)tp)[g * HIPK_TILE + t];
        unsigned w = 0xFFFFFFFFu;
        if (g == D) {
            const unsigned char *bp = tp + (size_t)D * 1024 + t;
            if (Bp >= 1) w = (w & 0xFFFFFF00u) | bp[0];
            if (Bp >= 2) w = (w & 0xFFFF00FFu) | ((unsigned)bp[HIPK_TILE] << 8);
        }
        return w;
    };
    constexpr int NE = UNITS == 0 ? 8 : (UNITS == 5 ? 5 : (UNITS == 4 ? 4 : 8));  // entries of the register groups
    struct req_t {  // what a tile needs before its x gathers can be issued
        unsigned c[G0];
        T w, b, d;
        T v[VALS ? NE : 1];  // VALS: the row's first NE values
        int D, Bp;
        const unsigned char *tp;
        const T *vp;  // VALS: this thread's element of the tile's value plane 0
    };
    const T *__restrict__ vals = (const T *)a.sell_vals;
    // UNI: uc = the tile's shared code bytes (0: read the planes), fetched one tile ahead of the request
    auto request = [&](int tl, req_t &q, unsigned long long uc) {  // tile tl's first G0 groups and epilogue operands
        const int r0 = tl * HIPK_TILE;
        if (UNITS > 0) {
            q.D = UNITS >> 2;
            q.Bp = UNITS & 3;
            q.tp = code + (size_t)tl * (UNITS * HIPK_TILE);
        } else {
            const int o0 = __builtin_amdgcn_readfirstlane(a.tile_off[tl]);
            const int o1 = __builtin_amdgcn_readfirstlane(a.tile_off[tl + 1]);
            q.D = (o1 - o0) >> 2;
            q.Bp = (o1 - o0) & 3;
            q.tp = code + (size_t)o0 * HIPK_TILE;
        }
        if (UNI && uc != 0ull) {
            q.c[0] = (unsigned)uc;
            if (G0 > 1) q.c[G0 - 1] = (unsigned)(uc >> 32);
        } else {
#pragma unroll
            for (int g = 0; g < G0; ++g) q.c[g] = load_group(q.tp, q.D, q.Bp, g);
        }
        if (VALS) {
            q.vp = vals + (size_t)(q.tp - code) + t;  // same prefix: a tile of U units has U value planes
            const int cap = 4 * q.D + q.Bp;
#pragma unroll
            for (int k = 0; k < NE; ++k) {
                q.v[k] = (T)0;
                if (UNITS > 0 || k < cap) q.v[k] = hipk_ld_nt(q.vp + (size_t)k * HIPK_TILE);
            }
        }
        q.w = (T)0;
        q.b = (T)0;
        q.d = (T)0;
        if (r0 + t < n32) {
            if (mode & HIPK_SPMV_DOT_W) q.w = ((const T *)a.w)[r0 + t];
            if (mode & HIPK_SPMV_RESID) q.b = ((const T *)a.bsub)[r0 + t];
            if (mode & HIPK_SPMV_SCALE) q.d = ((const T *)a.dscale)[r0 + t];
        }
    };
    auto gather = [&](const req_t &q, int tl, T(&xv)[NE]) {
        const int row = tl * HIPK_TILE + t;
        const int rowx = row < n32 ? row : n32 - 1;
#pragma unroll
        for (int k = 0; k < NE; ++k) {
            const unsigned ck = (q.c[k >> 2] >> ((k & 3) * 8)) & 0xFFu;
            const unsigned bo = (unsigned)(rowx + doff[ck]) * (unsigned)sizeof(T);
            xv[k] = *(const T *)(xb + bo);
        }
    };
    int cur = CHUNKED ? t_first : 0;
    auto first_tile = [&]() -> int {
        if (CHUNKED) return (t_first < t_end) ? t_first : ntiles;
        return (idx < per && xcd * per + idx < ntiles) ? xcd * per + idx : ntiles;
    };
    auto next_tile = [&]() -> int {  // this workgroup's next tile, ntiles when it has none left
        if (CHUNKED) {
            ++cur;
            return (cur < t_end) ? cur : ntiles;
        }
        idx += gp;
        return (idx < per && xcd * per + idx < ntiles) ? xcd * per + idx : ntiles;
    };

    // two tiles in flight per workgroup: the next tile's codes and epilogue operands travel while the current
    // tile's x gathers are outstanding (a third stage -- gathers one tile ahead -- measured no better)
    req_t rc, rn;
    T xc[NE];
    int tc = first_tile();
    const unsigned long long *__restrict__ ucode = a.tile_ucode;
    auto ldu = [&](int tl) -> unsigned long long {  // wave-uniform: kept in scalar registers
        const unsigned long long u = ucode[tl];
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u);
        const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
        return (unsigned long long)lo | ((unsigned long long)hi << 32);
    };
    unsigned long long uc = 0ull, un = 0ull;
    int tn = ntiles;
    if (UNI) {
        if (tc < ntiles) uc = ldu(tc);
        tn = next_tile();
        if (tn < ntiles) un = ldu(tn);
    }
    if (tc < ntiles) request(tc, rc, uc);
    if (a.stop_it != nullptr && a.it >= *a.stop_it) return;
    dval[t] = dv;  // slots >= n_codes, in particular HIPK_SELL_PAD: offset 0, value 0
    doff[t] = dofs;
    __syncthreads();

    while (tc < ntiles) {
        gather(rc, tc, xc);
        int t2 = ntiles;
        unsigned long long u2 = 0ull;
        if (UNI) {  // the tile after next: its shared code word is requested now, used by the next iteration's request
            t2 = next_tile();
            if (t2 < ntiles) u2 = ldu(t2);
        } else {
            tn = next_tile();
        }
        if (tn < ntiles) request(tn, rn, un);

        const int row = tc * HIPK_TILE + t;
        T s = (T)0;
#pragma unroll
        for (int k = 0; k < NE; ++k) {
            const unsigned ck = (rc.c[k >> 2] >> ((k & 3) * 8)) & 0xFFu;
            const T p = (VALS ? rc.v[k] : dval[ck]) * xc[k];
            const T s1 = s + p;
            s = (ck != HIPK_SELL_PAD) ? s1 : s;
        }
        if (UNITS == 0) {
            const int groups = rc.D + (rc.Bp > 0 ? 1 : 0);
            const int rowx = row < n32 ? row : n32 - 1;
            for (int g = G0; g < groups; ++g) {  // wider stencils: further groups of four codes
                const unsigned cw = load_group(rc.tp, rc.D, rc.Bp, g);
                const int cap = 4 * rc.D + rc.Bp;
                T xw[4], vw[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const unsigned ck = (cw >> (k * 8)) & 0xFFu;
                    const unsigned bo = (unsigned)(rowx + doff[ck]) * (unsigned)sizeof(T);
                    xw[k] = *(const T *)(xb + bo);
                    vw[k] = (T)0;
                    if (VALS && 4 * g + k < cap) vw[k] = hipk_ld_nt(rc.vp + (size_t)(4 * g + k) * HIPK_TILE);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const unsigned ck = (cw >> (k * 8)) & 0xFFu;
                    const T p = (VALS ? vw[k] : dval[ck]) * xw[k];
                    const T s1 = s + p;
                    s = (ck != HIPK_SELL_PAD) ? s1 : s;
                }
            }
        }
        double d0 = 0.0, d1 = 0.0;
        if (row < n32) {
            T out = s;
            if (mode & HIPK_SPMV_RESID) out = rc.b - out;
            if (mode & HIPK_SPMV_SCALE) out = rc.d * out;
            y[row] = out;
            if (mode & HIPK_SPMV_DOT_W) d0 = (double)rc.w * (double)out;
            if (mode & HIPK_SPMV_DOT_YY) d1 = (double)out * (double)out;
        }
        // fused dots: per-wavefront sums (shuffle tree 32..1) -> LDS (CHUNKED) or the tile-partial scratch
        const int slot = CHUNKED ? (tc - t_first) * 4 + wave : 0;
        const size_t tpi = (size_t)tc * 4 + wave;
        if (mode & HIPK_SPMV_DOT_W) {
            d0 = hipk_wave_sum(d0);
            if (lane == 0) {
                if (CHUNKED && gt == 0) wsum0[slot] = d0; else a.tpart0[tpi] = d0;
            }
        }
        if (mode & HIPK_SPMV_DOT_YY) {
            d1 = hipk_wave_sum(d1);
            if (lane == 0) {
                if (CHUNKED && gt == 0) wsum1[slot] = d1; else a.tpart1[tpi] = d1;
            }
        }
        rc = rn;
        tc = tn;
        if (UNI) {
            tn = t2;
            un = u2;
        }
    }
    if (CHUNKED && gt == 0 && (mode & (HIPK_SPMV_DOT_W | HIPK_SPMV_DOT_YY))) {
        __syncthreads();
        if (wave == 0) {  // the combine kernel's fold, on the LDS copy of this chunk's wavefront sums
            const int cnt = t_end - t_first;
            if (mode & HIPK_SPMV_DOT_W) {
                const double r = hipk_wave_fold(wsum0, cnt, lane);
                if (lane == 0) a.part0[chunk] = r;
            }
            if (mode & HIPK_SPMV_DOT_YY) {
                const double r = hipk_wave_fold(wsum1, cnt, lane);
                if (lane == 0) a.part1[chunk] = r;
            }
        }
    }
}
// ------------------------------------------------------------------ chunk per workgroup, TWO tiles per loop trip
// The persistent kernel above spends one x-gather round trip (~2.5 us at full occupancy) PER TILE per workgroup and about
// 250 executed instructions per tile and wavefront, a fifth of them register copies of the software pipeline
// (`rc = rn`) and hazard no-ops between the dependent DPP steps of the fused dot's wavefront sum: with one wave of
// workgroups (one reduction chunk = 8 tiles each at N = 4 M) the kernel lasts 8 round trips.  Here the workgroup takes its
// tiles in PAIRS: both tiles' gathers (2 x NE loads per thread) are in flight together, their row sums and wavefront
// sums are independent instruction streams the scheduler interleaves (no idle hazard slots), and the loop is unrolled over
// two pairs so that the prefetched requests change roles instead of being copied.  Same arithmetic per row, same tile /
// chunk folds: same bits.  Pair codes only (VALS = false), exact tile sizes (UNITS = 4, 5, 8), chunks of <= HIPK_SELL_MAX_TPC tiles.
// MODE >= 0: the mode bits as a compile-time constant (the solver loops' hot forms: no per-tile scalar branches on the mode)
template <typename T, int UNITS, bool UNI, int MODE = -1>
__global__ __launch_bounds__(HIPK_THREADS) void hipk_spmv_sell_pair_kernel(hipk_spmv_args a) {
    constexpr int G0 = (UNITS + 3) / 4;
    constexpr int NE = UNITS == 5 ? 5 : (UNITS == 4 ? 4 : 8);
    static_assert(UNITS == 4 || UNITS == 5 || UNITS == 8, "exact tile sizes only");
    const int ntiles = (int)((a.n + HIPK_TILE - 1) / HIPK_TILE);
    const int tpc = a.ch / HIPK_TILE;
    const int chunk = hipk_xcd_chunk(blockIdx.x, a.g);
    if (chunk < 0) return;
    const int t_first = chunk * tpc;
    const int t_end = (t_first + tpc < ntiles) ? t_first + tpc : ntiles;
    __shared__ double wsum0[HIPK_SELL_MAX_TPC * 4];
    __shared__ double wsum1[HIPK_SELL_MAX_TPC * 4];
    __shared__ T dval[HIPK_CODED_MAX];
    __shared__ int doff[HIPK_CODED_MAX];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const unsigned char *__restrict__ code = a.code;
    const char *__restrict__ xb = (const char *)a.x;
    T *__restrict__ y = (T *)a.y;
    const int mode = MODE >= 0 ? MODE : a.mode;
    const int n32 = (int)a.n;
    const unsigned long long *__restrict__ ucode = a.tile_ucode;

    T dv = (T)0;
    int dofs = 0;
    if (t < a.n_codes) {
        dv = ((const T *)a.dict_val)[t];
        dofs = a.dict_off[t];
    }
    struct req_t {  // the code groups of a tile: all that is fetched one trip ahead (8 VGPRs for the four requests in flight;
        unsigned c[G0];  // the epilogue operands travel with the x gathers, which keeps the kernel at 8 workgroups per CU)
    };
    struct ops_t {
        T w, b, d;
    };
    auto request = [&](int tl, req_t &q) {
        unsigned long long uc = 0ull;
        if (UNI) {
            const unsigned long long u = ucode[tl];  // wave-uniform: scalar load
            uc = (unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)u) |
                 ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(u >> 32)) << 32);
        }
        if (UNI && uc != 0ull) {
            q.c[0] = (unsigned)uc;
            if (G0 > 1) q.c[G0 - 1] = (unsigned)(uc >> 32);
        } else {
            const unsigned char *tp = code + (size_t)tl * (UNITS * HIPK_TILE);
            constexpr int D = UNITS >> 2, Bp = UNITS & 3;
#pragma unroll
            for (int g = 0; g < G0; ++g) {
                unsigned wv = 0xFFFFFFFFu;
                if (g < D) {
                    wv = ((const unsigned *)tp)[g * HIPK_TILE + t];
                } else {
                    const unsigned char *bp = tp + (size_t)D * 1024 + t;
                    if (Bp >= 1) wv = (wv & 0xFFFFFF00u) | bp[0];
                    if (Bp >= 2) wv = (wv & 0xFFFF00FFu) | ((unsigned)bp[HIPK_TILE] << 8);
                }
                q.c[g] = wv;
            }
        }
    };
    auto gather = [&](const req_t &q, int tl, T(&xv)[NE], ops_t &o) {
        const int row = tl * HIPK_TILE + t;
        const int rowx = row < n32 ? row : n32 - 1;
#pragma unroll
        for (int k = 0; k < NE; ++k) {
            const unsigned ck = (q.c[k >> 2] >> ((k & 3) * 8)) & 0xFFu;
            const unsigned bo = (unsigned)(rowx + doff[ck]) * (unsigned)sizeof(T);
            xv[k] = *(const T *)(xb + bo);
        }
        o.w = (T)0;
        o.b = (T)0;
        o.d = (T)0;
        if (row < n32) {
            if (mode & HIPK_SPMV_DOT_W) o.w = ((const T *)a.w)[row];
            if (mode & HIPK_SPMV_RESID) o.b = ((const T *)a.bsub)[row];
            if (mode & HIPK_SPMV_SCALE) o.d = ((const T *)a.dscale)[row];
        }
    };
    auto finish = [&](const req_t &q, int tl, const T(&xv)[NE], const ops_t &o) {
        const int row = tl * HIPK_TILE + t;
        T s = (T)0;
#pragma unroll
        for (int k = 0; k < NE; ++k) {
            const unsigned ck = (q.c[k >> 2] >> ((k & 3) * 8)) & 0xFFu;
            const T p = dval[ck] * xv[k];
            const T s1 = s + p;
            s = (ck != HIPK_SELL_PAD) ? s1 : s;
        }
        double d0 = 0.0, d1 = 0.0;
        if (row < n32) {
            T out = s;
            if (mode & HIPK_SPMV_RESID) out = o.b - out;
            if (mode & HIPK_SPMV_SCALE) out = o.d * out;
            y[row] = out;
            if (mode & HIPK_SPMV_DOT_W) d0 = (double)o.w * (double)out;
            if (mode & HIPK_SPMV_DOT_YY) d1 = (double)out * (double)out;
        }
        const int slot = (tl - t_first) * 4 + wave;
        if (mode & HIPK_SPMV_DOT_W) {
            d0 = hipk_wave_sum(d0);
            if (lane == 0) wsum0[slot] = d0;
        }
        if (mode & HIPK_SPMV_DOT_YY) {
            d1 = hipk_wave_sum(d1);
            if (lane == 0) wsum1[slot] = d1;
        }
    };
    // one trip: gathers of the pair (ca, cb) = tiles tp, tp + 1 | code requests of the next pair into (na, nb) | finish the pair
    auto trip = [&](req_t &ca, req_t &cb, req_t &na, req_t &nb, int tp) {
        const bool hb = tp + 1 < t_end;
        T xa[NE], xbv[NE];
        ops_t oa, ob;
        gather(ca, tp, xa, oa);
        if (hb) gather(cb, tp + 1, xbv, ob);
        if (tp + 2 < t_end) request(tp + 2, na);
        if (tp + 3 < t_end) request(tp + 3, nb);
        finish(ca, tp, xa, oa);
        if (hb) finish(cb, tp + 1, xbv, ob);
    };

    req_t r0, r1, r2, r3;
    if (t_first < t_end) request(t_first, r0);
    if (t_first + 1 < t_end) request(t_first + 1, r1);
    if (a.stop_it != nullptr && a.it >= *a.stop_it) return;
    dval[t] = dv;  // slots >= n_codes, in particular HIPK_SELL_PAD: offset 0, value 0
    doff[t] = dofs;
    __syncthreads();
    for (int tp = t_first; tp < t_end; tp += 4) {
        trip(r0, r1, r2, r3, tp);
        if (tp + 2 < t_end) trip(r2, r3, r0, r1, tp + 2);
    }
    if (mode & (HIPK_SPMV_DOT_W | HIPK_SPMV_DOT_YY)) {
        __syncthreads();
        if (wave == 0) {
            const int cnt = t_end - t_first;
            if (mode & HIPK_SPMV_DOT_W) {
                const double r = hipk_wave_fold(wsum0, cnt, lane);
                if (lane == 0) a.part0[chunk] = r;
            }
            if (mode & HIPK_SPMV_DOT_YY) {
                const double r = hipk_wave_fold(wsum1, cnt, lane);
                if (lane == 0) a.part1[chunk] = r;
            }
        }
    }
}
// ------------------------------------------------------------------ uniform tiles TWO ROWS PER LANE (16-byte accesses)
// tools/ubench/stencil_probe.hip took the kernels above apart on the N = 4 M 5-point matrix: a stencil kernel with NO per-lane
// decode at all (offsets and values in scalar registers, ~40 vector instructions per tile) runs no faster than the coded kernel
// (18.0 vs 18.4 us), the number of tiles in flight per trip does not matter (1, 2, 4, 8: 17.6-18.3 us), but every 8-byte-per-lane
// load instruction costs: 1 / 3 / 5 / 6 loads per row = 12.3 / 13.6 / 15.3 / 18.0 us, while the same five loads as 16-byte
// accesses (two adjacent rows per lane) cost 12.3 us -- what a plain copy of x to y costs in this launch shape.  The bound is the
// count of vector-memory instructions, not bytes, latency or vector ALU work.
// So: a UNIFORM tile (all 256 rows carry the same codes, `tile_ucode`) is taken by two wavefronts, each lane forming rows 2l and
// 2l + 1 of its wavefront's 128 rows from 16-byte loads at `x + offset_k` (the offset and value of code k live in scalar registers,
// fetched once per change of `ucode`); the other tiles (grid-line ends: about one in eight) take the per-lane path above, whole
// workgroup, first.  Wavefront pair p walks the uniform tiles of parity p of the chunk.  Same products, same additions in the same
// order per row; the fused dots' 64-row sums follow the spec's tree with the rows laid out two per lane (strides 32 ... 2 are lane
// shifts by 16 ... 1 on both components, stride 1 is x + y): same bits.  fp64, pair codes, chunk per workgroup.
__device__ __forceinline__ double hipk_half_tree2(double2 d) {  // sums of rows 0..63 / 64..127 of a wavefront: lanes 0 / 32
    d.x = d.x + hipk_lane_up16(d.x);
    d.y = d.y + hipk_lane_up16(d.y);
    d.x = d.x + hipk_row_shl<8>(d.x);
    d.y = d.y + hipk_row_shl<8>(d.y);
    d.x = d.x + hipk_row_shl<4>(d.x);
    d.y = d.y + hipk_row_shl<4>(d.y);
    d.x = d.x + hipk_row_shl<2>(d.x);
    d.y = d.y + hipk_row_shl<2>(d.y);
    d.x = d.x + hipk_row_shl<1>(d.x);
    d.y = d.y + hipk_row_shl<1>(d.y);
    return d.x + d.y;
}

// WALK = 1 (groups): the launch is not tied to the reduction chunks -- one workgroup per GROUP of 4 consecutive tiles on an ordinary
// grid, XCD k taking the k-th eighth of the groups; the wavefront sums of the fused dots go to the per-tile buffer and
// hipk_tile_combine_kernel folds them (same fold, same bits).  Two uses.
// (a) Row blocks with FEW chunks of many tiles (a rank of a row-partitioned system: 4 M rows in 245 chunks), which cannot fill the
//     chip with a workgroup per chunk and used to fall to the one-row-per-lane kernel.
// (b) Chunks of 64 tiles and more (N > 16 M on one device).  Counters of the N = 64 M matrix (profiles/r02_spmv_n64m_counters.md):
//     with a workgroup per chunk of 128 consecutive tiles every one of the three x streams of the stencil (rows -nx, 0, +nx) was
//     fetched from beyond L2 (FETCH_SIZE 1.58 GB per product for a 0.51 GB vector): a workgroup re-reads x[row +- nx] after 64 KB of
//     its own traffic, times the 244 workgroups of an XCD = 16 MB against 4 MB of L2.  A persistent grid with the workgroups of an
//     XCD statically interleaved over its eighth did no better (1.45 GB): memory-bound workgroups, unsynchronised over 61 trips,
//     drift apart by more tiles than that L2 holds.  Handing the tiles out by a per-XCD ticket counter proved the point (0.53 GB: x
//     once) and was 4.4 x SLOWER -- 125 k device-scope atomics on one line, 12 ns each.  The hardware dispatcher is the cheaper
//     ticket counter: it starts workgroups in ascending order as slots free up, so with short groups the tiles in flight in an XCD
//     are one compact front whatever the drift, and the re-reads of x[row +- nx] come from workgroups in flight at the same time
//     (the shape that works at N = 4 M, where a chunk IS 8 tiles): 0.53 GB fetched, 397 -> 273 us in the CG loop.
// Diagnostic twin (make stamps, -DHIPK_GM_STAMPS): every wavefront of the kernel below records the constant 100 MHz clock
// (s_memrealtime: comparable across compute units) at its phase boundaries; tools/spmv_stamps_probe.py prints where the time goes.
#ifdef HIPK_GM_STAMPS
#define HIPK_WIDE_NSTAMP 8
__device__ unsigned long long hipk_wide_stamps[2048 * 4 * HIPK_WIDE_NSTAMP];
#define HIPK_WSTAMP(k)                                                                                                   \
    do {                                                                                                                 \
        if (lane == 0 && blockIdx.x < 2048)                                                                              \
            hipk_wide_stamps[((size_t)blockIdx.x * 4 + wave) * HIPK_WIDE_NSTAMP + (k)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define HIPK_WSTAMP(k)
#endif
// HIPK_SGPR80 (hipk_common.h): the 8-wide instantiations and the run-time-mode ones sat at 82-106 scalar registers (seven or six
// workgroups per CU admitted); the stamps twin of the 5-wide one at 92 -- its first run showed workgroups 1792 .. 1953 starting 8 us
// late, an artefact of the twin (the product kernel has 68).
template <int UNITS, int MODE = -1, int WALK = 0>
__global__ __launch_bounds__(HIPK_THREADS) HIPK_SGPR80 void hipk_spmv_sell_wide_kernel(hipk_spmv_args a) {
    typedef double T;
    constexpr bool STRIDED = WALK != 0;  // tile sums to the per-tile buffer, no in-kernel fold
    constexpr int G0 = (UNITS + 3) / 4;
    constexpr int NE = UNITS == 5 ? 5 : (UNITS == 4 ? 4 : 8);
    static_assert(UNITS == 4 || UNITS == 5 || UNITS == 8, "exact tile sizes only");
    const int ntiles = (int)((a.n + HIPK_TILE - 1) / HIPK_TILE);
    const int tpc = a.ch / HIPK_TILE;
    const int chunk = STRIDED ? 0 : hipk_xcd_chunk(blockIdx.x, a.g);
    if (chunk < 0) return;
    // local tile i of this workgroup is tile t_first + i, i < cnt
    int t_first = chunk * tpc;
    int cnt = ((t_first + tpc < ntiles) ? t_first + tpc : ntiles) - t_first;
    if (WALK == 1) {
        const int ngroups = (ntiles + HIPK_SELL_GROUP - 1) / HIPK_SELL_GROUP;
        const int grp = hipk_xcd_chunk(blockIdx.x, ngroups);
        if (grp < 0) return;
        t_first = grp * HIPK_SELL_GROUP;
        cnt = (t_first + HIPK_SELL_GROUP < ntiles ? t_first + HIPK_SELL_GROUP : ntiles) - t_first;
    }
    __shared__ double wsum0[STRIDED ? 1 : HIPK_SELL_MAX_TPC * 4];
    __shared__ double wsum1[STRIDED ? 1 : HIPK_SELL_MAX_TPC * 4];
    __shared__ T dval[HIPK_CODED_MAX];
    __shared__ int doff[HIPK_CODED_MAX];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const unsigned char *__restrict__ code = a.code;
    const char *__restrict__ xb = (const char *)a.x;
    T *__restrict__ y = (T *)a.y;
    const int mode = MODE >= 0 ? MODE : a.mode;
    const int n32 = (int)a.n;
    const unsigned long long *__restrict__ ucode = a.tile_wcode;   // uniform and masked tiles (the launcher passes tile_ucode
    const int *__restrict__ g_doff = a.dict_off;                   // where there is no masked analysis)
    const T *__restrict__ g_dval = (const T *)a.dict_val;

    HIPK_WSTAMP(0);
    unsigned long long uc_mine = 0ull, uc_more = 0ull;  // requested first: complete before the dictionary reaches LDS (loads
    if (lane < cnt) uc_mine = ucode[t_first + lane];                                                 // return in order)
    if (!STRIDED && tpc > 64 && lane + 64 < cnt) uc_more = ucode[t_first + 64 + lane];
    T dv = (T)0;
    int dofs = 0;
    if (t < a.n_codes) {
        dv = g_dval[t];
        dofs = g_doff[t];
    }
    if (a.stop_it != nullptr && a.it >= *a.stop_it) return;
    dval[t] = dv;  // slots >= n_codes, in particular HIPK_SELL_PAD: offset 0, value 0
    doff[t] = dofs;
    HIPK_WSTAMP(1);
    __syncthreads();
    HIPK_WSTAMP(2);

    // the chunk's `ucode` words: lane i of every wavefront holds tile t_first + i's (and t_first + 64 + i's for chunks of more
    // than 64 tiles), a tile's word is read into scalar registers with v_readlane (no memory round trip per tile)
    auto tile_ucode_of = [&](int i) -> unsigned long long {  // of local tile i
        const unsigned long long src = (i & 64) ? uc_more : uc_mine;
        return (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)src, i & 63) |
               ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(src >> 32), i & 63) << 32);
    };

    // ---- a tile whose rows differ: one row per lane, codes from the planes (as hipk_spmv_sell_pair_kernel, one tile per trip);
    // whole workgroup.  i = the tile's local index (WALK 0: its slots of the LDS sums)
    auto per_lane_tile = [&](int tl, int i) {
        unsigned c[G0];
        {
            const unsigned char *tp = code + (size_t)tl * (UNITS * HIPK_TILE);
            constexpr int D = UNITS >> 2, Bp = UNITS & 3;
#pragma unroll
            for (int g = 0; g < G0; ++g) {
                unsigned wv = 0xFFFFFFFFu;
                if (g < D) {
                    wv = ((const unsigned *)tp)[g * HIPK_TILE + t];
                } else {
                    const unsigned char *bp = tp + (size_t)D * 1024 + t;
                    if (Bp >= 1) wv = (wv & 0xFFFFFF00u) | bp[0];
                    if (Bp >= 2) wv = (wv & 0xFFFF00FFu) | ((unsigned)bp[HIPK_TILE] << 8);
                }
                c[g] = wv;
            }
        }
        const int row = tl * HIPK_TILE + t;
        const int rowx = row < n32 ? row : n32 - 1;
        T xv[NE];
#pragma unroll
        for (int k = 0; k < NE; ++k) {
            const unsigned ck = (c[k >> 2] >> ((k & 3) * 8)) & 0xFFu;
            const unsigned bo = (unsigned)(rowx + doff[ck]) * (unsigned)sizeof(T);
            xv[k] = *(const T *)(xb + bo);
        }
        T ow = (T)0, ob = (T)0, od = (T)0;
        if (row < n32) {
            if (mode & HIPK_SPMV_DOT_W) ow = ((const T *)a.w)[row];
            if (mode & HIPK_SPMV_RESID) ob = ((const T *)a.bsub)[row];
            if (mode & HIPK_SPMV_SCALE) od = ((const T *)a.dscale)[row];
        }
        T s = (T)0;
#pragma unroll
        for (int k = 0; k < NE; ++k) {
            const unsigned ck = (c[k >> 2] >> ((k & 3) * 8)) & 0xFFu;
            const T p = dval[ck] * xv[k];
            const T s1 = s + p;
            s = (ck != HIPK_SELL_PAD) ? s1 : s;
        }
        double d0 = 0.0, d1 = 0.0;
        if (row < n32) {
            T out = s;
            if (mode & HIPK_SPMV_RESID) out = ob - out;
            if (mode & HIPK_SPMV_SCALE) out = od * out;
            y[row] = out;
            if (mode & HIPK_SPMV_DOT_W) d0 = (double)ow * (double)out;
            if (mode & HIPK_SPMV_DOT_YY) d1 = (double)out * (double)out;
        }
        const int slot = i * 4 + wave;
        if (mode & HIPK_SPMV_DOT_W) {
            d0 = hipk_wave_sum(d0);
            if (lane == 0) {
                if (STRIDED) a.tpart0[(size_t)tl * 4 + wave] = d0;
                else wsum0[slot] = d0;
            }
        }
        if (mode & HIPK_SPMV_DOT_YY) {
            d1 = hipk_wave_sum(d1);
            if (lane == 0) {
                if (STRIDED) a.tpart1[(size_t)tl * 4 + wave] = d1;
                else wsum1[slot] = d1;
            }
        }
    };

    // ---- a uniform tile: two rows per lane, one wavefront PAIR per tile (wh = this wavefront's half of the tile)
    const int wp = wave >> 1, wh = wave & 1;
    unsigned long long cur = 0ull;
    const bool w_is_x = (mode & HIPK_SPMV_DOT_W) && a.w == a.x;  // <x, A x> (the CG loop): w is the diagonal entry's operand
    int kc = -1;        // entry with offset 0, if any
    long long sbo[NE];  // byte offset of entry k (scalar)
    T sv[NE];           // its value (scalar)
#pragma unroll
    for (int k = 0; k < NE; ++k) {
        sbo[k] = 0;
        sv[k] = (T)0;
    }
    auto uniform_tile = [&](int tl, int i, unsigned long long uc_in) {
        // a MASKED tile: the rows have subsets of the pattern; the marker byte reads as padding below
        const bool masked = (unsigned)(uc_in >> 56) == HIPK_SELL_MASKED;
        const unsigned long long uc = masked ? (uc_in | 0xFF00000000000000ull) : uc_in;
        if (uc != cur) {
            cur = uc;
            kc = -1;
#pragma unroll
            for (int k = 0; k < NE; ++k) {
                const unsigned ck = (unsigned)(uc >> (8 * k)) & 0xFFu;
                const int ci = ck == HIPK_SELL_PAD ? 0 : (int)ck;
                sbo[k] = (long long)__builtin_amdgcn_readfirstlane(doff[ci]) * (long long)sizeof(T);
                if (ck != HIPK_SELL_PAD && sbo[k] == 0) kc = k;
                const double v = dval[ci];
                sv[k] = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)),
                                         __builtin_amdgcn_readfirstlane(__double2loint(v)));
            }
        }
        const int r0 = tl * HIPK_TILE + wh * 128 + 2 * lane;
        const unsigned vo = (unsigned)r0 * (unsigned)sizeof(T);
        double2 xv[NE];
#pragma unroll
        for (int k = 0; k < NE; ++k) {
            const unsigned ck = (unsigned)(uc >> (8 * k)) & 0xFFu;
            if (ck != HIPK_SELL_PAD) xv[k] = *(const double2 *)(xb + sbo[k] + vo);
        }
        double2 ow = {0.0, 0.0}, ob = {0.0, 0.0}, od = {0.0, 0.0};
        if (mode & HIPK_SPMV_DOT_W) {
            if (w_is_x && kc >= 0) {
#pragma unroll
                for (int k = 0; k < NE; ++k)
                    if (k == kc) ow = xv[k];
            } else {
                ow = *(const double2 *)((const char *)a.w + vo);
            }
        }
        if (mode & HIPK_SPMV_RESID) ob = *(const double2 *)((const char *)a.bsub + vo);
        if (mode & HIPK_SPMV_SCALE) od = *(const double2 *)((const char *)a.dscale + vo);
        unsigned pm = 0xFFFFu;   // presence of the pattern's entries in rows 2l (low byte) and 2l + 1 (high byte)
        if (masked) pm = *(const unsigned short *)(a.row_mask + r0);
        double2 s = {0.0, 0.0};
#pragma unroll
        for (int k = 0; k < NE; ++k) {
            const unsigned ck = (unsigned)(uc >> (8 * k)) & 0xFFu;
            if (ck != HIPK_SELL_PAD) {
                const double px = sv[k] * xv[k].x, py = sv[k] * xv[k].y;
                const double sx = s.x + px, sy = s.y + py;
                s.x = ((pm >> k) & 1u) ? sx : s.x;
                s.y = ((pm >> (8 + k)) & 1u) ? sy : s.y;
            }
        }
        double2 out = s;
        if (mode & HIPK_SPMV_RESID) {
            out.x = ob.x - out.x;
            out.y = ob.y - out.y;
        }
        if (mode & HIPK_SPMV_SCALE) {
            out.x = od.x * out.x;
            out.y = od.y * out.y;
        }
        *(double2 *)((char *)y + vo) = out;
        const int slot = i * 4 + 2 * wh + (lane >> 5);
        if (mode & HIPK_SPMV_DOT_W) {
            double2 d = {ow.x * out.x, ow.y * out.y};
            const double r = hipk_half_tree2(d);
            if ((lane & 31) == 0) {
                if (STRIDED) a.tpart0[(size_t)tl * 4 + 2 * wh + (lane >> 5)] = r;
                else wsum0[slot] = r;
            }
        }
        if (mode & HIPK_SPMV_DOT_YY) {
            double2 d = {out.x * out.x, out.y * out.y};
            const double r = hipk_half_tree2(d);
            if ((lane & 31) == 0) {
                if (STRIDED) a.tpart1[(size_t)tl * 4 + 2 * wh + (lane >> 5)] = r;
                else wsum1[slot] = r;
            }
        }
    };

    // ---- the tiles whose rows differ first, whole workgroup
    for (int i = 0; i < cnt; ++i) {
        if (tile_ucode_of(i) != 0ull) continue;
        per_lane_tile(t_first + i, i);
    }
    HIPK_WSTAMP(3);
    // ---- the uniform tiles of this wavefront pair's parity
    for (int i = wp; i < cnt; i += 2) {
        const unsigned long long uc = tile_ucode_of(i);
        if (uc == 0ull) continue;
        uniform_tile(t_first + i, i, uc);
#ifdef HIPK_GM_STAMPS
        if (i == wp) HIPK_WSTAMP(4);   // after the first uniform tile
#endif
    }
    HIPK_WSTAMP(5);
    if (!STRIDED && (mode & (HIPK_SPMV_DOT_W | HIPK_SPMV_DOT_YY))) {
        __syncthreads();
        HIPK_WSTAMP(6);
        if (wave == 0) {
            if (mode & HIPK_SPMV_DOT_W) {
                const double r = hipk_wave_fold(wsum0, cnt, lane);
                if (lane == 0) a.part0[chunk] = r;
            }
            if (mode & HIPK_SPMV_DOT_YY) {
                const double r = hipk_wave_fold(wsum1, cnt, lane);
                if (lane == 0) a.part1[chunk] = r;
            }
        }
    }
    HIPK_WSTAMP(7);
}
#endif
