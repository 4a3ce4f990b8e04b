// hipk_mid.h -- building blocks of the one-launch solver loops for mid-size systems (hipk_cg_mid.h, hipk_bi_mid.h): flagged
// 16-byte words for hand-offs between resident workgroups, the polled fold of chunk partials, one-barrier block folds, the
// matrix's reach beyond a reduction chunk.  See hipk_cg_mid.h for the scheme.
#ifndef HIPK_MID_H
#define HIPK_MID_H
#include "hipk_handoff.h"

static constexpr int kMidMinChunks = 8;       // up to 8 chunks the one-XCD kernel (hipk_cg_solve_lds_kernel, LOCAL) is faster
static constexpr int kMidMaxChunks = 512;     // two partials per thread in the fold
static constexpr size_t kMidSlotBytes = 2 * (size_t)kMidMaxChunks * 256;   // both partial arrays at the widest slot stride
static constexpr int kMidSpinBound = 1 << 18; // polls (~1 us each) before a workgroup gives up on a hand-off

// (A barrier that orders LDS traffic only -- s_waitcnt lgkmcnt(0); s_barrier instead of __syncthreads(), which also waits for the
// wavefront's write-through stores -- was measured after every publish of the three loops: no change; not kept.)
// One 16-byte store / load per flagged double: {lo, seq, hi, seq}.  Each 8-byte half validates itself, so a store or load torn
// into its halves is harmless.  Raw buffer accesses with the sc1 policy (aux bit 4) -- what the compiler gives relaxed agent-scope
// atomics: write-through stores, loads served at the device's coherence point -- so that they stay compiler-tracked: several polls
// of a thread are in flight together and a load can be issued long before its value is examined (the halo of r, below).
typedef unsigned hipk_v4u __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t hipk_ll_rsrc;
__device__ __forceinline__ hipk_ll_rsrc hipk_ll_make(const void *base, size_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, (int)bytes, 0x00020000);   // raw buffer, 32-bit data format
}
// (slot s of the array that starts `off` bytes into the resource: the voffset is s * 16, the soffset `off`)
__device__ __forceinline__ void hipk_ll_put(hipk_ll_rsrc rs, unsigned slot, double v, unsigned seq, unsigned off = 0) {
    const hipk_v4u w = {(unsigned)__double2loint(v), seq, (unsigned)__double2hiint(v), seq};
    __builtin_amdgcn_raw_buffer_store_b128(w, rs, slot * 16u, off, 16);
}
__device__ __forceinline__ hipk_v4u hipk_ll_load(hipk_ll_rsrc rs, unsigned slot, unsigned off = 0) {
    return __builtin_amdgcn_raw_buffer_load_b128(rs, slot * 16u, off, 16);
}
__device__ __forceinline__ bool hipk_ll_ok(const hipk_v4u w, unsigned seq) { return w.y == seq && w.w == seq; }
__device__ __forceinline__ double hipk_ll_val(const hipk_v4u w) { return __hiloint2double((int)w.z, (int)w.x); }
// poll one flagged word, starting from an earlier load's result; false when the spin bound was hit
// STG: keep a second load in flight while the first is examined -- a poll is a 0.5 us trip and a miss costs a whole one; with two
// loads a short sleep apart the word is seen sooner.  Library twins, same box, per iteration: CG 5.40 -> 5.18 us at 250 k rows,
// 10.3 -> 9.27 at 1 M; BiCGStab 4-6 % SLOWER (11.7 -> 12.3 .. 13.4 -> 13.8), GMRES unchanged -- taken by the CG loop only.
template <bool STG = false>
__device__ __forceinline__ bool hipk_ll_wait(hipk_ll_rsrc rs, unsigned slot, unsigned seq, hipk_v4u w, double &v, unsigned off = 0) {
    unsigned spins = 0;
    if (STG) {
        if (hipk_ll_ok(w, seq)) {
            v = hipk_ll_val(w);
            return true;
        }
        hipk_v4u w1 = hipk_ll_load(rs, slot, off);
        for (;;) {
            __builtin_amdgcn_s_sleep(2);
            const hipk_v4u w2 = hipk_ll_load(rs, slot, off);   // in flight while w1 is examined
            if (hipk_ll_ok(w1, seq)) {
                v = hipk_ll_val(w1);
                return true;
            }
            if (++spins > (unsigned)kMidSpinBound) return false;
            w1 = w2;
        }
    }
    while (!hipk_ll_ok(w, seq)) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (unsigned)kMidSpinBound) return false;
        w = hipk_ll_load(rs, slot, off);
    }
    v = hipk_ll_val(w);
    return true;
}

// ---- window plan: which columns a row block's window holds -------------------------------------------------------------------
// The LDS window of a workgroup is a list of 256-column TILES, ascending: every tile its rows reference plus its own.  For a 2-D
// stencil that is the contiguous range [base - nx, base + rows + nx); for a 3-D one three bands (own rows +- a grid line, and
// the planes below and above), a third of the contiguous range -- what makes 64^3 .. 80^3 grids fit.  Per row block b of
// `own` rows: tlo[b] (first tile of its range), nslot[b], tiles[b][s] (slot -> tile), map[b][tile - tlo] (tile -> slot or -1);
// needed[tile] = 1 when a block other than the owner holds the tile (its rows of r are published).  Computed once per handle.
static constexpr int kMidPlanRange = 512;   // tiles between the first and the last one a block references (131072 columns)
static constexpr int kMidPlanSlots = 64;    // tiles a window may hold (the LDS a kernel needs decides before this does)
struct hipk_mid_plan {
    const int *tlo, *nslot, *tiles;
    const short *map;
    const unsigned char *needed;
    int max_slots;
};
static __global__ __launch_bounds__(256) void hipk_mid_plan_kernel(const int *__restrict__ crow, const int *__restrict__ col, int64_t n,
                                                                   int own, int *__restrict__ tlo, int *__restrict__ nslot,
                                                                   int *__restrict__ tiles, short *__restrict__ map,
                                                                   unsigned char *__restrict__ needed, int *__restrict__ status) {
    __shared__ int tmin_s, tmax_s, flags[kMidPlanRange];
    const int b = blockIdx.x, t = threadIdx.x;
    const int64_t r0 = (int64_t)b * own, r1 = (r0 + own < n) ? r0 + own : n;
    const int own_lo = (int)(r0 >> 8), own_hi = (int)((r0 + own - 1) >> 8);   // the block's own tiles, all of them (also beyond n)
    if (t == 0) {
        tmin_s = own_lo;
        tmax_s = own_hi;
    }
    __syncthreads();
    int mn = own_lo, mx = own_hi;
    for (int64_t i = r0 + t; i < r1; i += 256)
        for (int e = crow[i]; e < crow[i + 1]; ++e) {
            const int tt = col[e] >> 8;
            mn = tt < mn ? tt : mn;
            mx = tt > mx ? tt : mx;
        }
    atomicMin(&tmin_s, mn);
    atomicMax(&tmax_s, mx);
    __syncthreads();
    const int tmin = tmin_s, L = tmax_s - tmin + 1;
    if (L > kMidPlanRange) {
        if (t == 0) {
            atomicOr(status, 1);
            nslot[b] = 0;
            tlo[b] = tmin;
        }
        return;
    }
    for (int i = t; i < kMidPlanRange; i += 256) flags[i] = (i + tmin >= own_lo && i + tmin <= own_hi) ? 1 : 0;
    __syncthreads();
    for (int64_t i = r0 + t; i < r1; i += 256)
        for (int e = crow[i]; e < crow[i + 1]; ++e) flags[(col[e] >> 8) - tmin] = 1;
    __syncthreads();
    if (t == 0) {
        int s = 0;
        for (int i = 0; i < L; ++i) {
            if (flags[i]) {
                if (s < kMidPlanSlots) tiles[b * kMidPlanSlots + s] = tmin + i;
                map[b * kMidPlanRange + i] = (short)s;
                ++s;
            } else {
                map[b * kMidPlanRange + i] = -1;
            }
        }
        nslot[b] = s;
        tlo[b] = tmin;
        if (s > kMidPlanSlots) atomicOr(status, 2);
        atomicMax(status + 1, s);
    }
    for (int i = t; i < L; i += 256)
        if (flags[i] && (i + tmin < own_lo || i + tmin > own_hi)) needed[i + tmin] = 1;
}
// the plan of handle A for row blocks of nch chunks (computed on first use; false: the matrix does not fit the scheme)
static inline bool hipk_mid_plan_get(hipk_csr_s *A, int nch, hipStream_t stream, hipk_mid_plan *out) {
    const int own = nch * HIPK_BASE_CHUNK;
    const int nblk = (int)((A->n_rows + own - 1) / own);
    const size_t ntiles_all = (size_t)nblk * (own / 256) + kMidPlanRange;
    const size_t o_tlo = 0, o_nslot = o_tlo + (size_t)nblk * 4, o_tiles = o_nslot + (size_t)nblk * 4,
                 o_map = o_tiles + (size_t)nblk * kMidPlanSlots * 4, o_needed = o_map + (size_t)nblk * kMidPlanRange * 2,
                 o_status = (o_needed + ntiles_all + 15) / 16 * 16, total = o_status + 16;
    if (A->mid_plan_state != 0 && A->mid_plan_nch != nch) {   // (a handle's chunk count never changes: one nch per handle)
        if (A->mid_plan_mem) (void)hipFree(A->mid_plan_mem);
        A->mid_plan_mem = nullptr;
        A->mid_plan_state = 0;
    }
    if (A->mid_plan_state == 0) {
        A->mid_plan_nch = nch;
        A->mid_plan_nblk = nblk;
        A->mid_plan_state = -1;
        int st[2] = {0, 0};
        if (hipMalloc(&A->mid_plan_mem, total) != hipSuccess) {
            A->mid_plan_mem = nullptr;
            (void)hipGetLastError();
            return false;
        }
        char *m = (char *)A->mid_plan_mem;
        bool ok = hipMemsetAsync(m, 0, total, stream) == hipSuccess;
        if (ok) {
            hipk_mid_plan_kernel<<<nblk, 256, 0, stream>>>(A->crow, A->col, A->n_rows, own, (int *)(m + o_tlo), (int *)(m + o_nslot),
                                                           (int *)(m + o_tiles), (short *)(m + o_map), (unsigned char *)(m + o_needed),
                                                           (int *)(m + o_status));
            ok = hipGetLastError() == hipSuccess && hipMemcpyAsync(st, m + o_status, sizeof(st), hipMemcpyDeviceToHost, stream) == hipSuccess &&
                 hipStreamSynchronize(stream) == hipSuccess;
        }
        if (ok && st[0] == 0) {
            A->mid_plan_state = 1;
            A->mid_plan_max_slots = st[1];
        }
    }
    if (A->mid_plan_state != 1) return false;
    char *m = (char *)A->mid_plan_mem;
    out->tlo = (const int *)(m + o_tlo);
    out->nslot = (const int *)(m + o_nslot);
    out->tiles = (const int *)(m + o_tiles);
    out->map = (const short *)(m + o_map);
    out->needed = (const unsigned char *)(m + o_needed);
    out->max_slots = A->mid_plan_max_slots;
    return true;
}

// thread t's share of the G flagged chunk partials in the spec's order (hipk_reduce_parts: t, t + 256; the tree follows);
// *fail set when a partial never arrived
template <int NK = kMidMaxChunks / 256, bool STG = false>
__device__ __forceinline__ double hipk_mid_poll(hipk_ll_rsrc rs, int g, unsigned seq, int *fail, int ss, unsigned off = 0) {
    const int t = threadIdx.x;
    hipk_v4u w[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k)
        if (t + k * 256 < g) w[k] = hipk_ll_load(rs, (t + k * 256) * ss, off);
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < NK; ++k)
        if (t + k * 256 < g) {
            double v = 0.0;
            if (!hipk_ll_wait<STG>(rs, (t + k * 256) * ss, seq, w[k], v, off)) *fail = 1;
            acc = acc + v;
        }
    return acc;
}

// the same for two / three slot arrays of ONE resource (byte offsets o0, o1, o2) at once: all loads go out before the first is examined
template <int NK = kMidMaxChunks / 256>
__device__ __forceinline__ void hipk_mid_poll3(hipk_ll_rsrc rs, unsigned o0, unsigned o1, unsigned o2, int g, unsigned seq, int *fail, int ss,
                                               double &a0, double &a1, double &a2, bool three = true) {
    const int t = threadIdx.x;   // NK: partials per thread (g <= 256 NK)
    hipk_v4u w0[NK], w1[NK], w2[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k)
        if (t + k * 256 < g) {
            w0[k] = hipk_ll_load(rs, (t + k * 256) * ss, o0);
            w1[k] = hipk_ll_load(rs, (t + k * 256) * ss, o1);
            if (three) w2[k] = hipk_ll_load(rs, (t + k * 256) * ss, o2);
        }
    a0 = a1 = a2 = 0.0;
#pragma unroll
    for (int k = 0; k < NK; ++k)
        if (t + k * 256 < g) {
            double v0 = 0.0, v1 = 0.0, v2 = 0.0;
            if (!hipk_ll_wait(rs, (t + k * 256) * ss, seq, w0[k], v0, o0)) *fail = 1;
            if (!hipk_ll_wait(rs, (t + k * 256) * ss, seq, w1[k], v1, o1)) *fail = 1;
            if (three && !hipk_ll_wait(rs, (t + k * 256) * ss, seq, w2[k], v2, o2)) *fail = 1;
            a0 = a0 + v0;
            a1 = a1 + v1;
            a2 = a2 + v2;
        }
}
template <int NK = kMidMaxChunks / 256>
__device__ __forceinline__ void hipk_mid_poll2(hipk_ll_rsrc rs, unsigned o0, unsigned o1, int g, unsigned seq, int *fail, int ss, double &a0,
                                               double &a1) {
    double a2;
    hipk_mid_poll3<NK>(rs, o0, o1, o1, g, seq, fail, ss, a0, a1, a2, false);
}

// The spec's fold of 256 per-thread values (hipk_block_sum: v[t] += v[t+128], v[t] += v[t+64], wavefront tree), done by EVERY
// wavefront for itself from the LDS copy sb[256]: no second barrier, no broadcast.  Same pairing, same bits.
__device__ __forceinline__ double hipk_mid_tree(const double *sb, int lane) {
    double a = (sb[lane] + sb[lane + 128]) + (sb[lane + 64] + sb[lane + 192]);
    a = hipk_wave_sum(a);
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(a)), __builtin_amdgcn_readfirstlane(__double2loint(a)));
}

// two wavefront sums of the tiled dot at once: lanes 0..31 take a[l] + a[l+32], lanes 32..63 b[l] + b[l+32] (one
// v_permlane32_swap per dword exchanges a's upper with b's lower half), then the strides 16 .. 1 run in both halves together.
// Sum of a in lane 0, of b in lane 32; the pairing -- and the bits -- of hipk_wave_sum on each.
__device__ __forceinline__ double hipk_wave_sum_pair(double a, double b) {
    const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(a), __double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(a), __double2hiint(b), false, false);
    const double x = __hiloint2double(hi[0], lo[0]), y = __hiloint2double(hi[1], lo[1]);
    return hipk_half_sum(x + y);
}

// the chunk's partial of a tiled dot from its 32 wavefront sums ts[32] (hipk_tile_combine_kernel's fold): lanes 0..7 of a
// wavefront take a tile each; valid in lane 0 (all 64 lanes must call)
__device__ __forceinline__ double hipk_mid_tiles_fold(const double *ts, int lane, int first_tile, int ntiles) {
    double tp = 0.0;
    if (lane < 8 && first_tile + lane < ntiles) tp = 0.0 + ((ts[lane * 4] + ts[lane * 4 + 1]) + (ts[lane * 4 + 2] + ts[lane * 4 + 3]));
    tp = tp + hipk_row_shl<4>(tp);   // (p0+p4) (p1+p5) (p2+p6) (p3+p7)
    tp = tp + hipk_row_shl<2>(tp);   // (p0+p4)+(p2+p6)  (p1+p5)+(p3+p7)
    tp = tp + hipk_row_shl<1>(tp);
    return 0.0 + tp;
}

#endif  // HIPK_MID_H
