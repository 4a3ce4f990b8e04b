// hipk_p2p.hip -- a peer-to-peer all-gather for the two tiny, latency-bound exchanges of a row-partitioned CG iteration.
//
// EXPERIMENTAL, opt-in (HIPK_DIST_COMM=p2p): built and tested with several ranks sharing ONE GPU (the only hardware this
// build has seen); its behaviour over xGMI between GPUs is unmeasured.  The default collective provider stays RCCL.
//
// Why: an iteration of the row-partitioned CG (hipk_dist.hip) makes two exchanges of a few KB per rank.  Through RCCL each
// is a collective launch of 15-25 us at 8 ranks -- as long as the iteration's three kernels together.  Here every rank owns a
// MAILBOX (device memory, exported with hipIpcGetMemHandle and mapped by every peer): 2 channels x world slots x max_count
// doubles + a flag word per slot.  One small kernel per exchange:
//   publish blocks (one per peer p): copy my `count` doubles into slot [channel][my_rank] of p's mailbox (system-scope stores),
//                                    fence, then store the call's sequence number into the slot's flag;
//   collect blocks (one per source s): spin (bounded) until MY mailbox's flag [channel][s] shows this call's sequence number,
//                                    fence, copy the slot into the caller's receive buffer.
// Channels alternate with the call number.  A peer can be at most ONE call ahead of me (it cannot finish call k+1 without my
// contribution to k+1, which I make after collecting k), so the slot a peer overwrites for call k+2 has long been read.
// All ranks must make the same sequence of calls with the same counts (the CG loop does).  The entry points have the
// signatures of hipk_rccl (include/hipk.h), so the C loop uses them unchanged; there is no send/recv (the loop then takes
// its all-gathered-slab form of the halo).
#include <stdlib.h>

#include "hipk_common.h"
#include "hipk_fx.h"
#include "hipk_solve.h"

#define HIPK_P2P_MAX_WORLD 64

struct hipk_p2p_s {
    int rank, world;
    size_t max_count;        // doubles per rank per call
    char *mine;              // my mailbox (device)
    size_t bytes;
    char *peer[HIPK_P2P_MAX_WORLD];   // every rank's mailbox as mapped into this process (peer[rank] == mine)
    char **peer_dev;         // device copy of `peer`
    unsigned long long seq;  // calls made so far
    int *err_dev;            // set by a collect block whose spin bound was hit
    bool uncached;
    // fused area (hipk_fx.h): exchanges folded into the CG kernels; fx_per == 0: none
    size_t fx_off;
    int fx_per, fx_ghost_cap;
    unsigned long long fx_seq;   // exchanges of each kind made so far (sequence numbers grow over the solves of a communicator)
};

static inline size_t hipk_p2p_flags_bytes(int world) { return hipk_align_up(sizeof(unsigned long long) * 2 * (size_t)world, 256); }
static inline size_t hipk_p2p_bytes(int world, size_t max_count) {
    return hipk_p2p_flags_bytes(world) + 2 * (size_t)world * hipk_align_up(max_count * sizeof(double), 256);
}

__global__ __launch_bounds__(HIPK_THREADS) void hipk_p2p_exchange_kernel(char *const *__restrict__ peer, int rank, int world,
                                                                         size_t slot_bytes, size_t flags_bytes, int ch,
                                                                         unsigned long long seq, const double *__restrict__ send,
                                                                         double *__restrict__ recv, size_t count, int *err) {
    const int b = blockIdx.x;
    if (b < world) {  // ---- publish to peer b
        char *box = peer[b];
        unsigned long long *dst = (unsigned long long *)(box + flags_bytes + ((size_t)ch * world + rank) * slot_bytes);
        for (size_t i = threadIdx.x; i < count; i += blockDim.x)
            __hip_atomic_store(dst + i, (unsigned long long)__double_as_longlong(send[i]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long *flag = (unsigned long long *)box + (size_t)ch * world + rank;
            __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    } else {          // ---- collect from source s
        const int s = b - world;
        char *box = peer[rank];
        __shared__ int ok;
        if (threadIdx.x == 0) {
            const unsigned long long *flag = (const unsigned long long *)box + (size_t)ch * world + s;
            unsigned spins = 0;
            ok = 1;
            while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1u << 22)) {  // seconds: the source never published -- report, do not hang
                    ok = 0;
                    atomicExch(err, 1);
                    break;
                }
            }
        }
        __syncthreads();
        __threadfence_system();
        const unsigned long long *src = (const unsigned long long *)(box + flags_bytes + ((size_t)ch * world + s) * slot_bytes);
        if (ok)
            for (size_t i = threadIdx.x; i < count; i += blockDim.x)
                recv[(size_t)s * count + i] = __longlong_as_double((long long)__hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
    }
}

extern "C" int hipk_p2p_create(hipk_p2p_t *out, int rank, int world, size_t max_count) {
    return hipk_p2p_create2(out, rank, world, max_count, 0, 0);
}

// fx_per > 0: the mailbox also carries the fused area of hipk_fx.h for a partition with `fx_per` chunks per rank and at most
// `fx_ghost_cap` ghost entries on any rank (both must be the same on every rank: the offsets inside a peer's mailbox follow)
extern "C" int hipk_p2p_create2(hipk_p2p_t *out, int rank, int world, size_t max_count, int fx_per, int fx_ghost_cap) {
    HIPK_REQUIRE(out && world >= 1 && world <= HIPK_P2P_MAX_WORLD && rank >= 0 && rank < world && max_count >= 1, HIPK_ERR_ARG,
                 "bad argument");
    HIPK_REQUIRE(fx_per >= 0 && fx_ghost_cap >= 0, HIPK_ERR_ARG, "bad fused-area geometry");
    hipk_p2p_s *c = new hipk_p2p_s();
    memset(c, 0, sizeof(*c));
    c->rank = rank;
    c->world = world;
    c->max_count = max_count;
    c->bytes = hipk_p2p_bytes(world, max_count);
    c->fx_off = c->bytes;
    c->fx_per = fx_per;
    c->fx_ghost_cap = fx_ghost_cap;
    if (fx_per > 0) c->bytes += hipk_fx_bytes(world, fx_per, fx_ghost_cap);
    void *p = nullptr;
    // uncached (fine-grained) device memory: peer stores become visible without cache maintenance on the owner
    hipError_t e = hipExtMallocWithFlags(&p, c->bytes, hipDeviceMallocUncached);
    c->uncached = (e == hipSuccess);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        e = hipMalloc(&p, c->bytes);
    }
    if (e == hipSuccess) e = hipMemset(p, 0, c->bytes);
    if (e == hipSuccess) e = hipMalloc((void **)&c->peer_dev, sizeof(char *) * HIPK_P2P_MAX_WORLD);
    if (e == hipSuccess) e = hipMalloc((void **)&c->err_dev, sizeof(int));
    if (e == hipSuccess) e = hipMemset(c->err_dev, 0, sizeof(int));
    if (e != hipSuccess) {
        hipk_set_error("hipk_p2p_create: %s", hipGetErrorString(e));
        if (p) (void)hipFree(p);
        delete c;
        return HIPK_ERR_HIP;
    }
    c->mine = (char *)p;
    c->peer[rank] = c->mine;
    *out = c;
    return HIPK_OK;
}

extern "C" int hipk_p2p_export(hipk_p2p_t c, void *handle64) {
    HIPK_REQUIRE(c && handle64, HIPK_ERR_ARG, "null argument");
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
    hipIpcMemHandle_t h;
    HIPK_CHECK_HIP(hipIpcGetMemHandle(&h, c->mine));
    memcpy(handle64, &h, 64);
    return HIPK_OK;
}

// handles: world x 64 bytes in rank order (this rank's own entry is ignored)
extern "C" int hipk_p2p_connect(hipk_p2p_t c, const void *handles) {
    HIPK_REQUIRE(c && handles, HIPK_ERR_ARG, "null argument");
    for (int r = 0; r < c->world; ++r) {
        if (r == c->rank) continue;
        hipIpcMemHandle_t h;
        memcpy(&h, (const char *)handles + 64 * (size_t)r, 64);
        void *p = nullptr;
        HIPK_CHECK_HIP(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
        c->peer[r] = (char *)p;
    }
    HIPK_CHECK_HIP(hipMemcpy(c->peer_dev, c->peer, sizeof(char *) * HIPK_P2P_MAX_WORLD, hipMemcpyHostToDevice));
    return HIPK_OK;
}

extern "C" int hipk_p2p_destroy(hipk_p2p_t c) {
    if (!c) return HIPK_OK;
    for (int r = 0; r < c->world; ++r)
        if (r != c->rank && c->peer[r]) (void)hipIpcCloseMemHandle(c->peer[r]);
    if (c->mine) (void)hipFree(c->mine);
    if (c->peer_dev) (void)hipFree(c->peer_dev);
    if (c->err_dev) (void)hipFree(c->err_dev);
    delete c;
    return HIPK_OK;
}

// 1 when a collect block gave up waiting since the LAST QUERY (the results of the calls in between are garbage).  Reading clears
// the flag (it used to stay set, so one timed-out exchange failed every later solve on the same mailboxes); the
// hipMemcpy / hipMemset pair runs on the null stream, i.e. after everything enqueued so far.
extern "C" int hipk_p2p_error(hipk_p2p_t c) {
    if (!c) return 0;
    int v = 0;
    if (hipMemcpy(&v, c->err_dev, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return 1;
    if (v != 0 && hipMemset(c->err_dev, 0, sizeof(int)) != hipSuccess) return 1;
    return v;
}

// ---- the hipk_rccl entry points (comm = hipk_p2p_t); return 0 on success like ncclResult_t
extern "C" int hipk_p2p_group_start(void) { return 0; }
extern "C" int hipk_p2p_group_end(void) { return 0; }
extern "C" int hipk_p2p_all_gather(const void *send, void *recv, size_t count, int datatype, void *comm, void *stream) {
    hipk_p2p_s *c = (hipk_p2p_s *)comm;
    if (!c || !send || !recv || datatype != 8 || count == 0 || count > c->max_count) return 4;  // ncclInvalidArgument
    const unsigned long long seq = ++c->seq;
    const int ch = (int)(seq & 1ull);
    hipk_p2p_exchange_kernel<<<2 * c->world, HIPK_THREADS, 0, (hipStream_t)stream>>>(
        c->peer_dev, c->rank, c->world, hipk_align_up(c->max_count * sizeof(double), 256), hipk_p2p_flags_bytes(c->world), ch, seq,
        (const double *)send, (double *)recv, count, c->err_dev);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}


// ---- fused exchanges (hipk_fx.h): what hipk_dist.hip needs from the communicator
// the constant part of the descriptor + the sequence number the solve's first exchange takes; 0 when the communicator has no fused
// area for this geometry or its mailbox is not fine-grained memory (peer stores must be visible without cache maintenance)
extern "C" int hipk_p2p_fx_begin(hipk_p2p_t c, int per, int n_ghost, hipk_fx *fx, unsigned long long *first_seq) {
    if (!c || !fx || !first_seq || c->fx_per <= 0 || c->fx_per != per || n_ghost > c->fx_ghost_cap || !c->uncached) return 0;
    memset(fx, 0, sizeof(*fx));
    fx->peer = c->peer_dev;
    fx->rank = c->rank;
    fx->world = c->world;
    fx->per = c->fx_per;
    fx->ghost_cap = c->fx_ghost_cap;
    fx->off_flags = c->fx_off;
    fx->off_parts = c->fx_off + hipk_fx_flags_bytes(c->world);
    fx->off_halo = fx->off_parts + hipk_fx_parts_bytes(c->world, c->fx_per);
    fx->err = c->err_dev;
    *first_seq = c->fx_seq + 1;
    return 1;
}
extern "C" void hipk_p2p_fx_end(hipk_p2p_t c, unsigned long long exchanges_issued) {
    if (c) c->fx_seq += exchanges_issued;
}
