// hipk_fx.h -- the two exchanges of a row-partitioned CG iteration FOLDED INTO ITS KERNELS.
//
// The row-partitioned CG (hipk_dist.hip) needs, per iteration, every rank's chunk partials of <p,Ap> before the update
// kernel and every rank's chunk partials of <r,r> plus the halo of r before the direction kernel.  Through RCCL these are two
// collective LAUNCHES per iteration (15-25 us each at 8 ranks: as long as the iteration's three kernels together); through the
// mailbox all-gather of hipk_p2p.hip still two small kernels.  Here they cost no launch at all:
//   * every rank's mailbox (hipk_p2p.hip: device memory, fine-grained, mapped by all peers through HIP IPC) carries a FUSED
//     AREA: per kind of exchange (0 = <p,Ap>, 1 = <r,r> + halo) and channel (iteration parity) one sequence flag per source
//     rank, a partials array laid out exactly like the global chunk-partial array the kernels fold (rank s owns entries
//     s * per ...), and -- kind 1 -- a halo array laid out exactly like the rank's ghost tail;
//   * the CONSUMER kernel of the exchange (update / direction, FX kernels in hipk_cg.hip) starts with: workgroups 0 .. world-1
//     PUBLISH -- workgroup q (q != this rank) stores this rank's partials (written by the previous kernel on the stream) and the
//     boundary entries of r that rank q's rows reference into q's mailbox with system-scope stores, fences, then stores the
//     exchange's sequence number into its flag there (the rank's own partials stay where they are: the collector reads them in place).  Workgroup 0 is also the COLLECTOR: it polls the `world` flags of its own mailbox
//     (one lane per source, bounded), folds the gathered partials -- the spec's fold, once -- and hands the SCALAR to the other
//     workgroups through a word of ordinary device memory (agent-scope store + sequence flag); they poll that flag and read
//     the scalar.  (First version: every workgroup polled the mailbox and folded from it -- 1954 workgroups x 31 KB of
//     fine-grained-memory reads cost 200 us per kernel at 4 M rows, 487 instead of 57 us per iteration at world 1.)
//     The workgroups whose chunk reaches into the ghost tail copy their part of the halo from the mailbox into r after the
//     flag.  Publishers and the collector never wait before they have published and have the lowest workgroup indices
//     (dispatched first), so the scheme cannot dead-lock on a grid of more workgroups than the chip holds.
// Ordering across iterations: a rank can be at most one exchange of a kind ahead of a peer (it cannot pass the wait of
// exchange k+1 without the peer's publication k+1, made after the peer consumed k), so two channels suffice.  Sequence
// numbers grow monotonically over the solves of a communicator.  Same partials, same fold, same bits as the single-device solve.
#pragma once
#include "hipk_common.h"

struct hipk_fx {
    char *const *peer;        // device array [world]: every rank's mailbox as mapped here (peer[rank] = my own)
    int rank, world, per, ghost_cap;
    size_t off_flags, off_parts, off_halo;  // byte offsets of the fused area's parts inside a mailbox (the same on every rank)
    unsigned long long seq;   // this exchange's number: a flag >= seq means "arrived"
    int ch, kind;             // channel = iteration parity; kind 0: <p,Ap> partials, 1: <r,r> partials + halo of `vec`
    int *err;                 // set when a wait gives up
    const double *parts;      // my `per` partials (previous kernel's output)
    const double *vec;        // kind 1: the vector whose boundary entries the peers need
    int n_ghost;
    // hand-off from the collector workgroup to the others: ordinary device memory (the solve's workspace)
    double *loc_val;              // [2 kinds][2 channels] folded scalars
    unsigned long long *loc_flag; // [2 kinds][8 replicas, 128 bytes apart]: sequence number of the last exchange whose scalar is in
                                  // loc_val; workgroup b polls replica b % 8 (one line per XCD's workgroups instead of one for all)
    const int *send_idx;      // device: my local rows grouped by destination rank
    const int *send_off;      // device [world + 1]: bounds of each destination's group in send_idx
    const long long *dest_off;  // device [world]: where my group starts in each destination's ghost tail
};

// layout of the fused area (host and device)
static inline size_t hipk_fx_flags_bytes(int world) { return ((sizeof(unsigned long long) * 4 * (size_t)world) + 255) / 256 * 256; }
static inline size_t hipk_fx_parts_bytes(int world, int per) { return ((sizeof(double) * 4 * (size_t)world * per) + 255) / 256 * 256; }
static inline size_t hipk_fx_halo_bytes(int ghost_cap) { return ((sizeof(double) * 2 * (size_t)(ghost_cap > 0 ? ghost_cap : 1)) + 255) / 256 * 256; }
static inline size_t hipk_fx_bytes(int world, int per, int ghost_cap) {
    return hipk_fx_flags_bytes(world) + hipk_fx_parts_bytes(world, per) + hipk_fx_halo_bytes(ghost_cap);
}

// hipk_p2p.hip
struct hipk_p2p_s;
extern "C" int hipk_p2p_fx_begin(hipk_p2p_s *c, int per, int n_ghost, hipk_fx *fx, unsigned long long *first_seq);
extern "C" void hipk_p2p_fx_end(hipk_p2p_s *c, unsigned long long exchanges_issued);
// hipk_cg.hip
int hipk_cg_update_fx(int64_t n_local, int chunk_rows, int g_red, const void *scal_dev, int64_t it, const void *Ap, void *r,
                      double *part_rr_out, const hipk_fx *fx, hipStream_t stream);
int hipk_cg_direction_fx(int64_t n_ext, int64_t n_local, int chunk_rows, int g_red, void *scal_dev, int64_t it, int64_t maxiter, void *r,
                         void *p, void *x, const hipk_fx *fx, hipStream_t stream);

#ifdef __HIPCC__
__device__ __forceinline__ void hipk_fx_store(double *p, double v) {
    __hip_atomic_store((unsigned long long *)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// the gathered partials of this exchange in MY mailbox: world * per doubles in global chunk order
__device__ __forceinline__ const double *hipk_fx_parts(const hipk_fx &fx) {
    return (const double *)(fx.peer[fx.rank] + fx.off_parts) + (size_t)(fx.kind * 2 + fx.ch) * fx.world * fx.per;
}
__device__ __forceinline__ const double *hipk_fx_parts_of(const hipk_fx &fx, int kind) {   // another kind of the same channel
    return (const double *)(fx.peer[fx.rank] + fx.off_parts) + (size_t)(kind * 2 + fx.ch) * fx.world * fx.per;
}
__device__ __forceinline__ const double *hipk_fx_halo(const hipk_fx &fx) {
    return (const double *)(fx.peer[fx.rank] + fx.off_halo) + (size_t)fx.ch * fx.ghost_cap;
}

// workgroups 0 .. world-1 (all their threads): publish to rank blockIdx.x.  Others: nothing.
__device__ __forceinline__ void hipk_fx_publish(const hipk_fx &fx) {
    const int q = blockIdx.x;
    if (q >= fx.world || q == fx.rank) return;   // this rank's own partials never leave its memory: the collector reads them in place
    char *box = fx.peer[q];
    double *dp = (double *)(box + fx.off_parts) + ((size_t)(fx.kind * 2 + fx.ch) * fx.world + fx.rank) * fx.per;
    for (int i = threadIdx.x; i < fx.per; i += blockDim.x) hipk_fx_store(dp + i, fx.parts[i]);
    if (fx.kind == 1 && q != fx.rank) {
        const int lo = fx.send_off[q], hi = fx.send_off[q + 1];
        double *hp = (double *)(box + fx.off_halo) + (size_t)fx.ch * fx.ghost_cap + fx.dest_off[q];
        for (int i = lo + (int)threadIdx.x; i < hi; i += blockDim.x) hipk_fx_store(hp + (i - lo), fx.vec[fx.send_idx[i]]);
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long *flag = (unsigned long long *)(box + fx.off_flags) + (size_t)(fx.kind * 2 + fx.ch) * fx.world + fx.rank;
        __hip_atomic_store(flag, fx.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// workgroup 0 (all its threads): wait until all `world` sources have published this exchange (lane s polls source s; bounded), fold
// the gathered partials with the spec's fold and publish the scalar to the other workgroups.  sbuf: HIPK_THREADS doubles.
__device__ __forceinline__ void hipk_fx_collect(const hipk_fx &fx, int g, double *sbuf, int *lds_ok) {
    if (threadIdx.x == 0) *lds_ok = 1;
    __syncthreads();
    if ((int)threadIdx.x < fx.world && (int)threadIdx.x != fx.rank) {
        const unsigned long long *flag =
            (const unsigned long long *)(fx.peer[fx.rank] + fx.off_flags) + (size_t)(fx.kind * 2 + fx.ch) * fx.world + threadIdx.x;
        unsigned spins = 0;
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < fx.seq) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 22)) {   // seconds: a source never published -- report, do not hang
                *lds_ok = 0;
                atomicExch(fx.err, 1);
                break;
            }
        }
    }
    __syncthreads();
    // hipk_reduce_parts over the gathered array -- thread t adds partials t, t + 256, ... in ascending order, then the block tree --
    // with this rank's own entries taken from where the previous kernel wrote them and the peers' from the mailbox (fine-grained
    // memory: those loads go to memory, after the polls above have returned; the only reads of the mailbox's partials on this rank)
    const double *mb = hipk_fx_parts(fx);
    const int own0 = fx.rank * fx.per;
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < HIPK_MAX_PARTS / HIPK_THREADS; ++k) {
        const int i = (int)threadIdx.x + k * HIPK_THREADS;
        if (i < g) acc = acc + ((i >= own0 && i < own0 + fx.per) ? fx.parts[i - own0] : mb[i]);
    }
    const double v = hipk_block_sum(acc, sbuf);
    if (threadIdx.x == 0) {
        __hip_atomic_store((unsigned long long *)&fx.loc_val[fx.kind * 2 + fx.ch], (unsigned long long)__double_as_longlong(v),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the scalar has left before the flag goes
        // a failed wait publishes the flag too (the others must not spin for ever); the error word voids the solve
    }
    if (threadIdx.x < 8) {
        __hip_atomic_store(&fx.loc_flag[(fx.kind * 8 + threadIdx.x) * 16], fx.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// every workgroup: wait for the collector's flag, return the folded scalar of `kind` (this exchange's or an earlier one of the
// same channel).  All threads must call it.
__device__ __forceinline__ void hipk_fx_await(const hipk_fx &fx) {
    if (threadIdx.x == 0) {
        unsigned spins = 0;
        // RELAXED agent-scope loads (as hipk_handoff.h's hand-offs): an ACQUIRE here made every poll of every workgroup
        // invalidate its L2 -- 605 us per iteration at 1954 workgroups.  The scalars are read with agent-scope loads as well,
        // the collector drains its stores (s_waitcnt) between the scalar and the flag: no fence is needed on this side.
        const unsigned long long *fl = &fx.loc_flag[(fx.kind * 8 + (blockIdx.x & 7)) * 16];
        while (__hip_atomic_load(fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < fx.seq) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 24)) {
                atomicExch(fx.err, 1);
                break;
            }
        }
    }
    __syncthreads();
}
__device__ __forceinline__ double hipk_fx_scalar(const hipk_fx &fx, int kind) {
    return __longlong_as_double((long long)__hip_atomic_load((const unsigned long long *)&fx.loc_val[kind * 2 + fx.ch], __ATOMIC_RELAXED,
                                                             __HIP_MEMORY_SCOPE_AGENT));
}
#endif
