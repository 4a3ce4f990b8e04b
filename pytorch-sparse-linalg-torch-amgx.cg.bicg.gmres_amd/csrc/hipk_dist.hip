// hipk_dist.hip -- row-partitioned CG, the per-iteration loop driven from C (one rank of it).
//
// New against the reference, which is single-process / single-device; the algorithm is `cg`
// (TSL:806-856 via `_isolve`, TSL:968-1016) on the SAME fused kernels as the single-GPU solve (include/hipk.h, step API).
// Between launches the ranks exchange, on the solver's stream, only
//   (1) the chunk partial sums of <p,Ap>                        : all-gather, <= 2048 doubles in total
//   (2) the chunk partial sums of <r,r> AND the halo entries of r: ONE group (all-gather + the halo, either neighbour
//       send/recv pairs or a second all-gather of padded slabs); every rank then forms the halo entries of
//       p = r + beta p and x += alpha p itself (same operands, same bits as the owner)
// so an iteration has two collective launches, the minimum for CG's two reductions.  Every rank folds the gathered
// partials in the same fixed order: the iterates are bitwise those of the single-GPU solve for any rank count.
//
// The loop used to be Python (pytorch_sparse_solver/distributed.py: 7 ctypes calls of ~4 us each per iteration -- host
// bound at 4 M rows per rank).  Here the host enqueues fixed BATCHES of iterations and learns the stop from a
// stream-ordered read of the device stop word one batch late (two reads in flight): all ranks derive the same stop word
// from the same gathered partials, so they take the same decision at the same batch boundary without talking to each
// other, and the number of collectives issued is identical on every rank (iterations past the stop are no-ops on the
// device; their collectives still pair up).  The RCCL entry points come in as function pointers resolved from the
// librccl that created the communicator (no link-time dependency; tests plug host-staged stand-ins in).
#include <stdlib.h>

#include <vector>

#include "hipk_common.h"
#include "hipk_fx.h"
#include "hipk_solve.h"

static inline size_t hipk_al(size_t v) { return hipk_align_up(v, 256); }

struct hipk_dist_layout {
    size_t scal, part_loc, spare, g_pAp, g_rr, g_bb, out4, fx_loc, send_buf, slab_loc, slab_all, p, r, Ap, total;
};

static hipk_dist_layout hipk_dist_make_layout(const hipk_dist_plan *pl) {
    hipk_dist_layout L;
    size_t o = 0;
    auto take = [&](size_t bytes) {
        const size_t at = o;
        o += hipk_al(bytes);
        return at;
    };
    const size_t per = (size_t)pl->per, W = (size_t)pl->world;
    const size_t next = (size_t)(pl->n_ext > 0 ? pl->n_ext : 1), nloc = (size_t)(pl->n_local > 0 ? pl->n_local : 1);
    L.scal = take(256);
    L.part_loc = take(per * 8);
    L.spare = take(per * 8);
    L.g_pAp = take(W * per * 8);
    L.g_rr = take(W * per * 8);
    L.g_bb = take(W * per * 8);
    L.out4 = take(4 * 8);
    L.fx_loc = take(256 + 2 * 8 * 128);   // fused exchanges: the collector's scalars [2][2] + sequence flags [2][8 replicas x 128 B]
    L.send_buf = take((size_t)(pl->n_send > 0 ? pl->n_send : 1) * 8);
    L.slab_loc = take((size_t)(pl->slab > 0 ? pl->slab : 1) * 8);
    L.slab_all = take((size_t)(pl->slab > 0 ? pl->slab : 1) * W * 8);
    L.p = take(next * 8);
    L.r = take(next * 8);
    L.Ap = take(nloc * 8);
    L.total = o;
    return L;
}

extern "C" size_t hipk_dist_cg_work_bytes(const hipk_dist_plan *plan) {
    if (!plan || plan->world < 1 || plan->per < 1) return 0;
    return hipk_dist_make_layout(plan).total;
}

#define HIPK_NCCL(expr, what)                                                        \
    do {                                                                             \
        const int _r = (expr);                                                       \
        if (_r != 0) {                                                               \
            hipk_set_error("hipk_dist_cg_solve: %s failed (ncclResult %d)", what, _r); \
            return HIPK_ERR_HIP;                                                     \
        }                                                                            \
    } while (0)
#define HIPK_TRY(expr)                  \
    do {                                \
        const int _rc = (expr);         \
        if (_rc != HIPK_OK) return _rc; \
    } while (0)

extern "C" int hipk_dist_cg_solve(hipk_csr_t A, const hipk_dist_plan *pl, const hipk_rccl *cc, const void *b_local,
                                  void *x_ext, void *work, size_t work_bytes, const hipk_params *prm, hipk_stats *st,
                                  hipk_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    HIPK_REQUIRE(A && pl && cc && b_local && x_ext && work && prm && st, HIPK_ERR_ARG, "null argument");
    HIPK_REQUIRE(A->dtype == HIPK_F64, HIPK_ERR_UNSUPPORTED, "the row-partitioned solver is fp64");
    HIPK_REQUIRE(pl->world >= 1 && pl->rank >= 0 && pl->rank < pl->world, HIPK_ERR_ARG, "rank / world");
    HIPK_REQUIRE(pl->n_local > 0 && pl->n_local == A->n_rows && pl->n_ext >= pl->n_local, HIPK_ERR_ARG,
                 "every rank must own rows (n_local > 0) and n_ext >= n_local");
    HIPK_REQUIRE(pl->per >= 1 && (int64_t)pl->per * pl->world >= pl->g_red && pl->g_red >= 1 && pl->g_red <= HIPK_MAX_PARTS,
                 HIPK_ERR_ARG, "partial-sum geometry");
    HIPK_REQUIRE((pl->n_local + pl->chunk_rows - 1) / pl->chunk_rows <= pl->per, HIPK_ERR_ARG, "more local chunks than `per`");
    HIPK_REQUIRE(cc->all_gather && cc->group_start && cc->group_end && (pl->world == 1 || pl->halo_mode == 0 || (cc->send && cc->recv)),
                 HIPK_ERR_ARG, "missing collective entry points");
    HIPK_REQUIRE((((uintptr_t)work) & 255u) == 0 && hipk_aligned16(x_ext) && hipk_aligned16(b_local), HIPK_ERR_ALIGN,
                 "work must be 256-byte, x / b 16-byte aligned");
    const hipk_dist_layout L = hipk_dist_make_layout(pl);
    HIPK_REQUIRE(work_bytes >= L.total, HIPK_ERR_WORKSPACE, "work too small");
    memset(st, 0, sizeof(*st));

    char *wk = (char *)work;
    void *scal = wk + L.scal;
    double *part_loc = (double *)(wk + L.part_loc), *spare = (double *)(wk + L.spare);
    double *g_pAp = (double *)(wk + L.g_pAp), *g_rr = (double *)(wk + L.g_rr), *g_bb = (double *)(wk + L.g_bb);
    double *out4 = (double *)(wk + L.out4);
    double *send_buf = (double *)(wk + L.send_buf), *slab_loc = (double *)(wk + L.slab_loc), *slab_all = (double *)(wk + L.slab_all);
    double *p = (double *)(wk + L.p), *r = (double *)(wk + L.r), *Ap = (double *)(wk + L.Ap);
    double *x = (double *)x_ext;
    const double *b = (const double *)b_local;
    const int64_t n = pl->n_local, n_ext = pl->n_ext;
    const int ch = pl->chunk_rows, G = pl->g_red, per = pl->per, W = pl->world;
    const int64_t *stop_dev = (const int64_t *)((char *)scal + 48);   // hipk_cg_scal::stop_it
    const int64_t maxiter = (prm->maxiter < 0) ? 10 * pl->n_global : prm->maxiter;   // TSL:982-984
    const int NCCL_F64 = 8;
    enum { MODE_DOT_W = 1, MODE_DOT_YY = 2, MODE_RESID = 4 };

    hipk_event_pair whole;
    HIPK_CHECK_HIP(whole.create());
    HIPK_CHECK_HIP(hipEventRecord(whole.a, stream));
    HIPK_CHECK_HIP(hipMemsetAsync(wk, 0, L.p, stream));                                   // scalars, partial slots, pack buffers
    HIPK_CHECK_HIP(hipMemsetAsync(p, 0, (size_t)n_ext * 8, stream));
    HIPK_CHECK_HIP(hipMemsetAsync(r, 0, (size_t)n_ext * 8, stream));
    if (n_ext > n) HIPK_CHECK_HIP(hipMemsetAsync(x + n, 0, (size_t)(n_ext - n) * 8, stream));

    auto gather_parts = [&](double *dst) -> int {
        HIPK_NCCL(cc->all_gather(part_loc, dst, (size_t)per, NCCL_F64, cc->comm, stream), "all_gather(partials)");
        return HIPK_OK;
    };
    // the peers' entries this rank's rows reference -> v[n .. n_ext)   (neighbour send/recv pairs, one group)
    // contiguous send ranges go out straight from the vector; the pack kernel runs only if some peer's list is scattered
    bool need_pack = false;
    for (int peer = 0; peer < W; ++peer)
        if (pl->send_counts[peer] > 0 && !(pl->send_first && pl->send_first[peer] >= 0)) need_pack = true;
    auto halo_p2p_calls = [&](double *v) -> int {
        size_t so = 0, ro = 0;
        for (int peer = 0; peer < W; ++peer) {
            const size_t ns = (size_t)pl->send_counts[peer], nr = (size_t)pl->recv_counts[peer];
            const bool direct = pl->send_first && pl->send_first[peer] >= 0;
            if (ns) HIPK_NCCL(cc->send(direct ? v + pl->send_first[peer] : send_buf + so, ns, NCCL_F64, peer, cc->comm, stream), "send(halo)");
            if (nr) HIPK_NCCL(cc->recv(v + n + ro, nr, NCCL_F64, peer, cc->comm, stream), "recv(halo)");
            so += ns;
            ro += nr;
        }
        return HIPK_OK;
    };
    auto halo_exchange = [&](double *v) -> int {   // stand-alone form (setup and the final residual)
        if (W == 1 || (pl->n_send == 0 && pl->n_ghost == 0 && pl->halo_mode == 1)) return HIPK_OK;
        if (pl->halo_mode == 1) {
            if (pl->n_send && need_pack) HIPK_TRY(hipk_gather(pl->n_send, pl->send_idx_dev, v, send_buf, HIPK_F64, stream));
            HIPK_NCCL(cc->group_start(), "group_start");
            HIPK_TRY(halo_p2p_calls(v));
            HIPK_NCCL(cc->group_end(), "group_end");
        } else {
            if (pl->n_send) HIPK_TRY(hipk_gather(pl->n_send, pl->send_idx_dev, v, slab_loc, HIPK_F64, stream));
            HIPK_NCCL(cc->all_gather(slab_loc, slab_all, (size_t)pl->slab, NCCL_F64, cc->comm, stream), "all_gather(halo slabs)");
            if (pl->n_ghost) HIPK_TRY(hipk_gather(pl->n_ghost, pl->ghost_src_dev, slab_all, v + n, HIPK_F64, stream));
        }
        return HIPK_OK;
    };
    // partials of <r,r> and the halo of r in ONE group
    auto gather_parts_and_halo = [&](double *dst, double *v) -> int {
        if (W == 1) return gather_parts(dst);
        if (pl->halo_mode == 1) {
            if (pl->n_send && need_pack) HIPK_TRY(hipk_gather(pl->n_send, pl->send_idx_dev, v, send_buf, HIPK_F64, stream));
            HIPK_NCCL(cc->group_start(), "group_start");
            HIPK_NCCL(cc->all_gather(part_loc, dst, (size_t)per, NCCL_F64, cc->comm, stream), "all_gather(partials)");
            HIPK_TRY(halo_p2p_calls(v));
            HIPK_NCCL(cc->group_end(), "group_end");
        } else {
            if (pl->n_send) HIPK_TRY(hipk_gather(pl->n_send, pl->send_idx_dev, v, slab_loc, HIPK_F64, stream));
            HIPK_NCCL(cc->group_start(), "group_start");
            HIPK_NCCL(cc->all_gather(part_loc, dst, (size_t)per, NCCL_F64, cc->comm, stream), "all_gather(partials)");
            HIPK_NCCL(cc->all_gather(slab_loc, slab_all, (size_t)pl->slab, NCCL_F64, cc->comm, stream), "all_gather(halo slabs)");
            HIPK_NCCL(cc->group_end(), "group_end");
            if (pl->n_ghost) HIPK_TRY(hipk_gather(pl->n_ghost, pl->ghost_src_dev, slab_all, v + n, HIPK_F64, stream));
        }
        return HIPK_OK;
    };

    // ---- r0 = b - A x0, <r0,r0>; <b,b> (TSL:815-826); halos of x and r0 explicitly once
    HIPK_TRY(halo_exchange(x));
    HIPK_TRY(hipk_spmv_ex(A, x, r, MODE_RESID | MODE_DOT_YY, nullptr, b, spare, part_loc, nullptr, 0, stream));
    HIPK_TRY(gather_parts(g_rr));
    HIPK_TRY(hipk_dot_parts(n, ch, b, b, HIPK_F64, part_loc, stream));
    HIPK_TRY(gather_parts(g_bb));
    HIPK_TRY(halo_exchange(r));
    HIPK_TRY(hipk_cg_start(n, ch, G, scal, g_rr, g_bb, r, p, HIPK_F64, prm->tol, prm->atol, maxiter, stream));
    if (n_ext > n)   // p0 = r0 on the halo as well
        HIPK_CHECK_HIP(hipMemcpyAsync(p + n, r + n, (size_t)(n_ext - n) * 8, hipMemcpyDeviceToDevice, stream));

    // ---- the loop: fixed batches, the stop word read one batch late (two reads in flight)
    int64_t batch = prm->check_every > 0 ? prm->check_every : 16;
    hipk_poller poll(A->host_poll);
    HIPK_CHECK_HIP(poll.create());
    int64_t it = 0, stop = INT64_MAX;
    // HIPK_DIST_OVERLAP=1 (opt-in): x += alpha p leaves the direction kernel and runs on a SIDE STREAM while the second collective
    // of the iteration (the <r,r> partials + the halo of r) is in flight -- it needs alpha (first collective) and the old p only;
    // the direction kernel then streams 24 n instead of 40 n bytes behind the collective.  Same operands, same bits.  Not the
    // default: the two cross-stream event dependencies per iteration cost more than the 9 us they can hide (world-1 rehearsal
    // at 4 M rows: 98 us per iteration against 66 fused).
    const char *ov_env = getenv("HIPK_DIST_OVERLAP");
    const bool overlap = ov_env ? atoi(ov_env) != 0 : false;
    hipStream_t side = nullptr;
    hipEvent_t ev_upd = nullptr, ev_x = nullptr;
    if (overlap) {
        HIPK_CHECK_HIP(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
        HIPK_CHECK_HIP(hipEventCreateWithFlags(&ev_upd, hipEventDisableTiming));
        HIPK_CHECK_HIP(hipEventCreateWithFlags(&ev_x, hipEventDisableTiming));
    }
    struct side_guard {   // every exit path releases the side stream and its events
        hipStream_t &s;
        hipEvent_t &a, &b;
        ~side_guard() {
            if (s) {
                (void)hipStreamSynchronize(s);
                (void)hipStreamDestroy(s);
            }
            if (a) (void)hipEventDestroy(a);
            if (b) (void)hipEventDestroy(b);
        }
    } side_release{side, ev_upd, ev_x};
    // FUSED exchanges (hipk_fx.h): with a mailbox communicator that carries a fused area (hipk_rccl.fused) the two exchanges of an
    // iteration are made by the update / direction kernels themselves -- three launches per iteration, as on one device.
    // HIPK_DIST_FUSED=0 keeps the collective entry points.
    hipk_fx fx;
    unsigned long long fx_seq0 = 0;
    const char *fx_env = getenv("HIPK_DIST_FUSED");
    bool fused = cc->fused != nullptr && !(fx_env && fx_env[0] == '0') && !overlap && (W == 1 || (pl->send_off_dev && pl->dest_off_dev)) &&
                 hipk_p2p_fx_begin((hipk_p2p_s *)cc->fused, per, pl->n_ghost, &fx, &fx_seq0) != 0;
    if (fused) {
        fx.send_idx = pl->send_idx_dev;
        fx.send_off = pl->send_off_dev;
        fx.dest_off = (const long long *)pl->dest_off_dev;
        fx.n_ghost = pl->n_ghost;
        fx.loc_val = (double *)(wk + L.fx_loc);
        fx.loc_flag = (unsigned long long *)(wk + L.fx_loc + 256);
        // (zeroed with the rest of the workspace header above: below every sequence number, which start at 1 and only grow)
    }
    struct fx_guard {   // every exit path tells the communicator how many exchanges were issued
        const hipk_rccl *cc;
        const bool &on;
        const int64_t &it;
        ~fx_guard() {
            if (on) hipk_p2p_fx_end((hipk_p2p_s *)cc->fused, (unsigned long long)it);
        }
    } fx_release{cc, fused, it};
    while (it < maxiter) {
        const int64_t end = (it + batch < maxiter) ? it + batch : maxiter;
        for (; fused && it < end; ++it) {
            HIPK_TRY(hipk_spmv_ex(A, p, Ap, MODE_DOT_W, p, nullptr, part_loc, spare, stop_dev, it, stream));
            fx.seq = fx_seq0 + (unsigned long long)it;
            fx.ch = (int)(it & 1);
            fx.kind = 0;
            fx.parts = part_loc;     // <p,Ap> partials of this rank's chunks (the SpMV's)
            fx.vec = nullptr;
            HIPK_TRY(hipk_cg_update_fx(n, ch, G, scal, it, Ap, r, spare, &fx, stream));
            fx.kind = 1;
            fx.parts = spare;        // <r,r> partials (the update kernel's)
            fx.vec = r;
            HIPK_TRY(hipk_cg_direction_fx(n_ext, n, ch, G, scal, it, maxiter, r, p, x, &fx, stream));
        }
        for (; it < end; ++it) {
            HIPK_TRY(hipk_spmv_ex(A, p, Ap, MODE_DOT_W, p, nullptr, part_loc, spare, stop_dev, it, stream));
            HIPK_TRY(gather_parts(g_pAp));
            HIPK_TRY(hipk_cg_update(n, ch, G, scal, it, g_pAp, Ap, r, part_loc, HIPK_F64, stream));
            if (overlap) {
                HIPK_CHECK_HIP(hipEventRecord(ev_upd, stream));
                HIPK_CHECK_HIP(hipStreamWaitEvent(side, ev_upd, 0));
                HIPK_TRY(hipk_cg_xupdate(n_ext, ch, G, scal, it, g_pAp, p, x, HIPK_F64, side));
                HIPK_CHECK_HIP(hipEventRecord(ev_x, side));
            }
            HIPK_TRY(gather_parts_and_halo(g_rr, r));
            if (overlap) HIPK_CHECK_HIP(hipStreamWaitEvent(stream, ev_x, 0));   // p is about to be replaced
            HIPK_TRY(hipk_cg_direction(n_ext, ch, G, scal, it, maxiter, g_pAp, g_rr, r, p, overlap ? nullptr : x, HIPK_F64, stream));
        }
        // every rank posts and harvests at the same points: the decision below is a function of values all ranks share
        HIPK_CHECK_HIP(poll.post(stop_dev, it, stream));
        if (poll.count == 2) {
            HIPK_CHECK_HIP(hipEventSynchronize(poll.ev[poll.head]));
            poll.harvest(&stop);
        }
        if (stop <= it - batch) break;   // the batch BEFORE the one just enqueued had already reached the stop
    }
    HIPK_CHECK_HIP(poll.drain(&stop));
    const int64_t iterations = stop < it ? stop : it;

    // ---- TSL:1007-1014: true residual and ||x|| decide info
    HIPK_TRY(halo_exchange(x));
    HIPK_TRY(hipk_spmv_ex(A, x, Ap, MODE_RESID | MODE_DOT_YY, nullptr, b, spare, part_loc, nullptr, 0, stream));
    HIPK_TRY(gather_parts(g_rr));
    HIPK_TRY(hipk_reduce_parts(g_rr, G, out4 + 0, stream));
    HIPK_TRY(hipk_dot_parts(n, ch, x, x, HIPK_F64, part_loc, stream));
    HIPK_TRY(gather_parts(g_pAp));
    HIPK_TRY(hipk_reduce_parts(g_pAp, G, out4 + 1, stream));
    HIPK_TRY(hipk_reduce_parts(g_bb, G, out4 + 2, stream));
    double h4[4] = {0, 0, 0, 0};
    HIPK_CHECK_HIP(hipEventRecord(whole.b, stream));
    HIPK_CHECK_HIP(hipMemcpyAsync(h4, out4, sizeof(h4), hipMemcpyDeviceToHost, stream));
    HIPK_CHECK_HIP(hipStreamSynchronize(stream));
    hipk_finish_isolve_stats(st, prm, h4[2], h4[0], h4[1], iterations, iterations + 2);
    float ms = 0.f;
    HIPK_CHECK_HIP(hipEventElapsedTime(&ms, whole.a, whole.b));
    st->solve_ms = ms;
    return HIPK_OK;
}
