"""Row-partitioned multi-GPU CG: one process per GPU, RCCL (torch.distributed 'nccl') over xGMI.

New against the reference, which is single-process/single-device.  The
solver is the SAME algorithm as `cg` (TSL:806-856 via `_isolve`, TSL:968-1016) and runs the
SAME fused gfx950 kernels as the single-GPU path (include/hipk.h "step API"); between
launches the ranks exchange only

  * the x-vector HALO before each SpMV: the entries of p a rank's rows reference but do not
    own (2 x ny doubles for the 5-point stencil), `all_to_all_single` of packed slabs --
    never the full vector;
  * the chunk PARTIAL SUMS of each dot (<= 2048 doubles in total): `all_gather_into_tensor`,
    after which EVERY rank folds all partials in the same fixed order.  Rows are split on
    reduction-chunk boundaries of the global problem, so the result is bitwise identical
    to the single-GPU solve for any number of ranks (tests/test_distributed_gloo.py).

The orchestration below is backend-agnostic (`ops` object): the product ships `HipOps`
(libhipk.so on CUDA/ROCm tensors); the CPU test double lives in tests/ and is never
imported from here.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import Optional

import torch
import torch.distributed as dist

BASE_CHUNK = 2048
MAX_PARTS = 2048
MODE_DOT_W, MODE_DOT_YY, MODE_RESID = 1, 2, 4


def chunk_geometry(n: int):
    """(chunk size, chunk count) of an n-vector: hipk_make_geom (csrc/hipk_common.h) in Python."""
    full = BASE_CHUNK * MAX_PARTS
    q = max(1, (n + full - 1) // full)
    p = 1
    while p < q:
        p <<= 1
    ch = BASE_CHUNK * p
    return ch, max(1, (n + ch - 1) // ch)


@dataclass
class RowPartition:
    """Contiguous row blocks aligned to the reduction chunks of the global problem."""
    n_global: int
    world: int
    rank: int
    force_ch: int = 0      # tests / experiments: emulate the chunk size of a larger global problem (2048 * 2^k)

    def __post_init__(self):
        self.ch, self.g = chunk_geometry(self.n_global)
        if self.force_ch:
            assert self.force_ch >= self.ch and self.force_ch % BASE_CHUNK == 0
            self.ch = self.force_ch
            self.g = max(1, (self.n_global + self.ch - 1) // self.ch)
        self.per = (self.g + self.world - 1) // self.world          # chunks per rank (last ranks may have fewer)
        self.c0 = min(self.g, self.rank * self.per)
        self.c1 = min(self.g, (self.rank + 1) * self.per)
        self.row0 = min(self.n_global, self.c0 * self.ch)
        self.row1 = min(self.n_global, self.c1 * self.ch)
        self.n_local = self.row1 - self.row0
        self.g_local = self.c1 - self.c0

    def owner_of(self, rows: torch.Tensor) -> torch.Tensor:
        return torch.clamp(torch.div(rows, self.ch * self.per, rounding_mode="floor"), max=self.world - 1)

    def bounds(self, r: int):
        c0, c1 = min(self.g, r * self.per), min(self.g, (r + 1) * self.per)
        return min(self.n_global, c0 * self.ch), min(self.n_global, c1 * self.ch)


class HaloPlan:
    """Which owned entries each peer needs from me, and where received entries land (built once)."""

    def __init__(self, col_global: torch.Tensor, part: RowPartition, group=None):
        dev = col_global.device
        owned = (col_global >= part.row0) & (col_global < part.row1)
        ghosts = torch.unique(col_global[~owned])                     # sorted global ids I need
        self.n_ghost = int(ghosts.numel())
        owners = part.owner_of(ghosts)
        self.recv_splits = torch.bincount(owners, minlength=part.world).tolist() if self.n_ghost else [0] * part.world
        # local numbering: owned -> [0, n_local), ghost k -> n_local + k
        col_local = torch.empty_like(col_global)
        col_local[owned] = col_global[owned] - part.row0
        if self.n_ghost:
            col_local[~owned] = part.n_local + torch.searchsorted(ghosts, col_global[~owned])
        self.col_local = col_local
        # tell every owner which of its rows I need (setup-time collective)
        send_counts = torch.tensor(self.recv_splits, dtype=torch.int64, device=dev)
        recv_counts = torch.empty_like(send_counts)
        dist.all_to_all_single(recv_counts, send_counts, group=group)
        self.send_splits = recv_counts.tolist()
        wanted = torch.empty(int(sum(self.send_splits)), dtype=torch.int64, device=dev)
        dist.all_to_all_single(wanted, ghosts.to(torch.int64), self.send_splits, self.recv_splits, group=group)
        self.send_idx = (wanted - part.row0).to(torch.int32)          # my local rows, grouped by destination
        self.n_send = int(self.send_idx.numel())
        if self.n_send:
            assert int(self.send_idx.min()) >= 0 and int(self.send_idx.max()) < part.n_local
        # per destination: first local row if its list is one contiguous ascending range (row blocks of a stencil), else -1
        self.send_first, o = [], 0
        idx_h = self.send_idx.cpu()
        for cnt in self.send_splits:
            seg = idx_h[o:o + cnt]
            ok = cnt > 0 and bool((seg[1:] - seg[:-1] == 1).all())
            self.send_first.append(int(seg[0]) if ok else -1)
            o += cnt
        # all-gather form of the same exchange (the halo can then ride with another all-gather):
        # every rank contributes its packed send list padded to `slab` entries; ghost k of owner o sits at
        # o*slab + (offset of my block in o's send list) + (k's index in my request to o)
        t = torch.tensor([self.n_send], dtype=torch.int64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        self.slab = max(int(t.item()), 1)
        my_off = torch.cumsum(torch.tensor([0] + self.send_splits[:-1], dtype=torch.int64, device=dev), 0)
        their_off = torch.empty_like(my_off)                          # their_off[o] = where my block starts in o's list
        dist.all_to_all_single(their_off, my_off, group=group)
        # fused exchanges (csrc/hipk_fx.h): every owner stores my ghost entries straight into my ghost tail, so it must know
        # where its group starts there -- and I, where mine starts in each destination's tail; the largest tail bounds the mailboxes
        ghost_first = torch.cumsum(torch.tensor([0] + self.recv_splits[:-1], dtype=torch.int64, device=dev), 0)
        dest_off = torch.empty_like(ghost_first)
        dist.all_to_all_single(dest_off, ghost_first, group=group)
        self.dest_off = dest_off.to(torch.int64)                       # [world]
        self.send_off = torch.cumsum(torch.tensor([0] + self.send_splits, dtype=torch.int64, device=dev), 0).to(torch.int32)
        t = torch.tensor([self.n_ghost], dtype=torch.int64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        self.ghost_cap = max(int(t.item()), 1)
        if self.n_ghost:
            first = torch.cumsum(torch.tensor([0] + self.recv_splits[:-1], dtype=torch.int64, device=dev), 0)
            idx_in_req = torch.arange(self.n_ghost, device=dev) - first[owners]
            self.ghost_src = (owners * self.slab + their_off[owners] + idx_in_req).to(torch.int32)
        else:
            self.ghost_src = torch.zeros(0, dtype=torch.int32, device=dev)


class HipOps:
    """The product backend: libhipk.so kernels on this rank's GPU (no CPU fallback)."""

    def __init__(self, device):
        from . import _hipk
        self.k = _hipk
        self.L = _hipk.lib()
        self.device = torch.device(device)
        self.dtype = torch.float64

    # -- memory
    def empty(self, n, dtype=None):
        return torch.empty(n, dtype=dtype or self.dtype, device=self.device)

    def zeros(self, n, dtype=None):
        return torch.zeros(n, dtype=dtype or self.dtype, device=self.device)

    def _s(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def _p(self, t):
        return None if t is None else t.data_ptr()

    def make_matrix(self, crow, col_local, val, n_local, n_cols_ext, ch):
        h = ctypes.c_void_p()
        crow, col_local, val = crow.contiguous(), col_local.to(crow.dtype).contiguous(), val.contiguous()
        with torch.cuda.device(self.device):
            rc = self.L.hipk_csr_create_ex(ctypes.byref(h), n_local, n_cols_ext, val.numel(), crow.data_ptr(),
                                           col_local.data_ptr(), crow.element_size(), val.data_ptr(),
                                           self.k.HIPK_F64, ch, self._s())
        self.k._check(rc, "hipk_csr_create_ex")
        return {"h": h, "keep": (crow, col_local, val)}

    def free_matrix(self, m):
        self.L.hipk_csr_destroy(m["h"])

    def spmv(self, m, x_ext, y, mode=0, w=None, bsub=None, part0=None, part1=None, stop=None, it=0):
        self.k._check(self.L.hipk_spmv_ex(m["h"], x_ext.data_ptr(), y.data_ptr(), mode, self._p(w), self._p(bsub),
                                          self._p(part0), self._p(part1), self._p(stop), it, self._s()), "hipk_spmv_ex")

    def dot_parts(self, n, ch, x, y, part):
        self.k._check(self.L.hipk_dot_parts(n, ch, x.data_ptr(), y.data_ptr(), self.k.HIPK_F64, part.data_ptr(),
                                            self._s()), "hipk_dot_parts")

    def reduce_parts(self, part, g):
        out = self.empty(1)
        self.k._check(self.L.hipk_reduce_parts(part.data_ptr(), g, out.data_ptr(), self._s()), "hipk_reduce_parts")
        return out

    def gather(self, idx, src, dst):
        self.k._check(self.L.hipk_gather(idx.numel(), idx.data_ptr(), src.data_ptr(), dst.data_ptr(), self.k.HIPK_F64,
                                         self._s()), "hipk_gather")

    def scal_alloc(self):
        return torch.zeros(int(self.L.hipk_cg_scal_bytes()) // 8, dtype=torch.float64, device=self.device)

    def stop_word(self, scal):
        return scal[6:7].view(torch.int64)          # hipk_cg_scal.stop_it

    def cg_start(self, n, ch, g, scal, part_rr, part_bb, r, p, tol, atol, maxiter):
        self.k._check(self.L.hipk_cg_start(n, ch, g, scal.data_ptr(), part_rr.data_ptr(), part_bb.data_ptr(),
                                           r.data_ptr(), p.data_ptr(), self.k.HIPK_F64, float(tol), float(atol),
                                           maxiter, self._s()), "hipk_cg_start")

    def cg_update(self, n, ch, g, scal, it, part_pAp, Ap, r, part_out):
        self.k._check(self.L.hipk_cg_update(n, ch, g, scal.data_ptr(), it, part_pAp.data_ptr(), Ap.data_ptr(),
                                            r.data_ptr(), part_out.data_ptr(), self.k.HIPK_F64, self._s()),
                      "hipk_cg_update")

    def cg_direction(self, n, ch, g, scal, it, maxiter, part_pAp, part_rr, r, p, x):
        self.k._check(self.L.hipk_cg_direction(n, ch, g, scal.data_ptr(), it, maxiter, part_pAp.data_ptr(),
                                               part_rr.data_ptr(), r.data_ptr(), p.data_ptr(), x.data_ptr(),
                                               self.k.HIPK_F64, self._s()), "hipk_cg_direction")

    def read_scal(self, scal):
        h = scal.cpu()
        return {"gamma": (h[0].item(), h[1].item()), "atol2": h[2].item(), "bs": h[3].item(),
                "stop_it": int(h[6:7].view(torch.int64).item())}


class RcclComm:
    """Direct RCCL communicator (ctypes on torch's own librccl.so) for the per-iteration collectives.

    torch.distributed costs 20-40 us of host time per collective; three per CG iteration made the
    Python-driven loop host-bound (152 vs 118 us/iteration at world 1).  Calling ncclAllGather /
    ncclSend / ncclRecv directly on the solver's stream costs a few microseconds and needs no
    cross-stream events.  The unique id is distributed once through torch.distributed."""
    DOUBLE = 8  # ncclFloat64 (rccl.h)

    class _Uid(ctypes.Structure):
        _fields_ = [("internal", ctypes.c_char * 128)]

    def __init__(self, rank: int, world: int, device, group=None):
        import os
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        self.L = L = ctypes.CDLL(path)
        vp, sz, i32 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
        L.ncclGetErrorString.restype = ctypes.c_char_p
        L.ncclGetUniqueId.argtypes = [ctypes.POINTER(self._Uid)]
        L.ncclCommInitRank.argtypes = [ctypes.POINTER(vp), i32, self._Uid, i32]
        L.ncclCommDestroy.argtypes = [vp]
        L.ncclAllGather.argtypes = [vp, vp, sz, i32, vp, vp]
        L.ncclSend.argtypes = [vp, sz, i32, i32, vp, vp]
        L.ncclRecv.argtypes = [vp, sz, i32, i32, vp, vp]
        self.rank, self.world, self.device = rank, world, torch.device(device)
        uid = self._Uid()
        if rank == 0:
            self._ck(L.ncclGetUniqueId(ctypes.byref(uid)), "ncclGetUniqueId")
        # raw 128 bytes: a c_char array FIELD reads back truncated at the first NUL
        box = [ctypes.string_at(ctypes.byref(uid), 128) if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(box, src=0, group=group)
        assert len(box[0]) == 128
        ctypes.memmove(ctypes.byref(uid), box[0], 128)
        self.comm = vp()
        with torch.cuda.device(self.device):
            self._ck(L.ncclCommInitRank(ctypes.byref(self.comm), world, uid, rank), "ncclCommInitRank")

    def _ck(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed: {self.L.ncclGetErrorString(rc).decode()}")

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def all_gather(self, dst: torch.Tensor, src: torch.Tensor) -> None:
        assert dst.numel() == src.numel() * self.world and dst.dtype == src.dtype == torch.float64
        self._ck(self.L.ncclAllGather(src.data_ptr(), dst.data_ptr(), src.numel(), self.DOUBLE, self.comm,
                                      self._stream()), "ncclAllGather")

    def all_gather2(self, dst_a, src_a, dst_b, src_b) -> None:
        """Two all-gathers in one RCCL group (aggregated into a single launch)."""
        s = self._stream()
        self._ck(self.L.ncclGroupStart(), "ncclGroupStart")
        self._ck(self.L.ncclAllGather(src_a.data_ptr(), dst_a.data_ptr(), src_a.numel(), self.DOUBLE, self.comm, s),
                 "ncclAllGather")
        self._ck(self.L.ncclAllGather(src_b.data_ptr(), dst_b.data_ptr(), src_b.numel(), self.DOUBLE, self.comm, s),
                 "ncclAllGather")
        self._ck(self.L.ncclGroupEnd(), "ncclGroupEnd")

    def all_to_all(self, recv: torch.Tensor, send: torch.Tensor, recv_splits, send_splits) -> None:
        """Grouped ncclSend/ncclRecv of contiguous slabs (doubles), zero-length pairs skipped."""
        s = self._stream()
        self._ck(self.L.ncclGroupStart(), "ncclGroupStart")
        so = ro = 0
        for peer in range(self.world):
            ns, nr = send_splits[peer], recv_splits[peer]
            if ns:
                self._ck(self.L.ncclSend(send.data_ptr() + 8 * so, ns, self.DOUBLE, peer, self.comm, s), "ncclSend")
            if nr:
                self._ck(self.L.ncclRecv(recv.data_ptr() + 8 * ro, nr, self.DOUBLE, peer, self.comm, s), "ncclRecv")
            so += ns
            ro += nr
        self._ck(self.L.ncclGroupEnd(), "ncclGroupEnd")

    def coll_struct(self):
        """hipk_rccl (include/hipk.h): the entry points of THIS librccl + the communicator, for the C-driven loop."""
        from . import _hipk
        addr = lambda f: ctypes.cast(f, ctypes.c_void_p).value   # noqa: E731
        return _hipk.Rccl(addr(self.L.ncclGroupStart), addr(self.L.ncclGroupEnd), addr(self.L.ncclAllGather),
                          addr(self.L.ncclSend), addr(self.L.ncclRecv), self.comm.value, None)

    def close(self):
        if getattr(self, "comm", None):
            self.L.ncclCommDestroy(self.comm)
            self.comm = None


class P2PComm:
    """EXPERIMENTAL exchange provider (csrc/hipk_p2p.hip, HIPK_DIST_COMM=p2p): every rank's device mailbox is mapped by
    its peers through HIP IPC and an all-gather is one small kernel per rank instead of an RCCL collective launch.
    Only the C-driven loop uses it (all-gathered-slab halo); the IPC handles travel once through torch.distributed."""

    def __init__(self, rank: int, world: int, device, max_count: int, group=None, fx_per: int = 0, fx_ghost_cap: int = 0):
        from . import _hipk
        self.L = L = _hipk.lib()
        self.rank, self.world, self.device = rank, world, torch.device(device)
        self.ctx = ctypes.c_void_p()
        self.fused = fx_per > 0     # the mailbox carries the fused area (csrc/hipk_fx.h): exchanges made by the CG kernels themselves
        with torch.cuda.device(self.device):
            _hipk._check(L.hipk_p2p_create2(ctypes.byref(self.ctx), rank, world, max(int(max_count), 1), int(fx_per), int(fx_ghost_cap)),
                         "hipk_p2p_create2")
            mine = ctypes.create_string_buffer(64)
            _hipk._check(L.hipk_p2p_export(self.ctx, mine), "hipk_p2p_export")
            box = [None] * world
            if world > 1:
                dist.all_gather_object(box, mine.raw, group=group)
            else:
                box[0] = mine.raw
            assert all(len(h) == 64 for h in box)
            _hipk._check(L.hipk_p2p_connect(self.ctx, b"".join(box)), "hipk_p2p_connect")
        if world > 1:
            dist.barrier(group=group)   # every mailbox is mapped before anyone publishes

    def coll_struct(self):
        from . import _hipk
        addr = lambda f: ctypes.cast(f, ctypes.c_void_p).value   # noqa: E731
        return _hipk.Rccl(addr(self.L.hipk_p2p_group_start), addr(self.L.hipk_p2p_group_end),
                          addr(self.L.hipk_p2p_all_gather), None, None, self.ctx.value, self.ctx.value if self.fused else None)

    def failed(self) -> bool:
        return bool(self.L.hipk_p2p_error(self.ctx))

    def close(self):
        if getattr(self, "ctx", None):
            self.L.hipk_p2p_destroy(self.ctx)
            self.ctx = None


@dataclass
class DistStats:
    iterations: int
    matvecs: int
    info: int
    b_norm: float
    residual_norm: float
    x_norm: float
    threshold: float
    method: str = ""


class DistProblem:
    """A rank's row block of a global CSR system, its halo plan and its device matrix."""

    def __init__(self, crow: torch.Tensor, col_global: torch.Tensor, val: torch.Tensor, b_local: torch.Tensor,
                 part: RowPartition, ops, group=None):
        self.part, self.ops, self.group = part, ops, group
        assert crow.numel() == part.n_local + 1 and b_local.numel() == part.n_local
        self.plan = HaloPlan(col_global, part, group)
        self.n_local = part.n_local
        self.n_ext = part.n_local + self.plan.n_ghost
        self.nnz_local = int(val.numel())
        self.b = b_local.contiguous()
        self.A = ops.make_matrix(crow, self.plan.col_local, val, part.n_local, max(self.n_ext, 1), part.ch) \
            if part.n_local > 0 else None
        # algorithmic bytes of this rank's SpMV (SURVEY 8d formula on the local block)
        self.spmv_bytes = self.nnz_local * 12 + (part.n_local + 1) * 4 + 2 * part.n_local * 8
        self.send_buf = ops.empty(max(self.plan.n_send, 1))
        self.slab_loc = ops.zeros(self.plan.slab)
        self.slab_all = ops.zeros(self.plan.slab * part.world)
        # per-iteration collectives: direct RCCL when the ranks are GPUs of an 'nccl' group, else torch.distributed
        self.comm = None
        self.comm_kind = "torch.distributed"
        import os
        want = os.environ.get("HIPK_DIST_COMM", "rccl")
        if want == "rccl" and isinstance(ops, HipOps) and dist.is_initialized() and dist.get_backend(group) == "nccl":
            # The choice between the direct communicator (C-driven loops) and torch.distributed collectives (Python loop) is
            # made HERE, once, by all ranks together and before any solve: each rank creates its communicator and runs one
            # tiny all-gather through it, then the ranks agree (MIN over a torch.distributed all-reduce) -- a rank whose
            # communicator failed must not leave its peers inside an RCCL call of a solve.  After this point an
            # error return of hipk_dist_*_solve on one rank is fatal for the group: the rank raises, its process exits
            # non-zero and the launcher (torch.distributed.run) tears the job down.
            ok, why = 1, ""
            try:
                self.comm = RcclComm(part.rank, part.world, ops.device, group)
                probe_src = torch.full((1,), float(part.rank), dtype=torch.float64, device=ops.device)
                probe_dst = torch.empty(part.world, dtype=torch.float64, device=ops.device)
                self.comm.all_gather(probe_dst, probe_src)
                torch.cuda.synchronize(ops.device)
                if probe_dst.tolist() != [float(r) for r in range(part.world)]:
                    raise RuntimeError(f"probe all-gather returned {probe_dst.tolist()}")
            except Exception as e:  # noqa: BLE001
                ok, why = 0, str(e)
            if part.world > 1:
                flag = torch.tensor([ok], dtype=torch.int32, device=ops.device)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
                all_ok = int(flag.item())
            else:
                all_ok = ok
            if all_ok:
                self.comm_kind = "rccl-direct"
            else:  # stay functional on the well-trodden torch.distributed path, on EVERY rank
                import warnings
                if self.comm is not None:
                    self.comm.close()
                self.comm = None
                self.comm_kind = "torch.distributed (the direct RCCL communicator failed on a rank)"
                warnings.warn(f"direct RCCL communicator unavailable on a rank ({why or 'another rank'}); "
                              f"all ranks use torch.distributed collectives")
        # opt-in: device mailboxes instead of RCCL for the C-driven loop's two exchanges (experimental, see P2PComm)
        # "fused": the mailboxes also carry the fused area -- the two exchanges of a CG iteration are made by its update /
        # direction kernels themselves, no collective launch inside the loop (csrc/hipk_fx.h); set-up and the final residual go
        # through the mailbox all-gather.  Built and tested with ranks sharing one GPU; over xGMI unmeasured, so RCCL stays default.
        self.p2p = None
        if want in ("p2p", "fused") and isinstance(ops, HipOps) and (part.world == 1 or dist.is_initialized()):
            fx = want == "fused"
            self.p2p = P2PComm(part.rank, part.world, ops.device, max(part.per, self.plan.slab), group,
                               fx_per=part.per if fx else 0, fx_ghost_cap=self.plan.ghost_cap if fx else 0)
            self.comm_kind = "p2p-mailbox, exchanges fused into the CG kernels" if fx else "p2p-mailbox"

    def coll_struct(self):
        """hipk_rccl for the C-driven loop, or None (then the Python loop with torch.distributed collectives runs)."""
        if getattr(self, "p2p", None) is not None:
            return self.p2p.coll_struct()
        return self.comm.coll_struct() if self.comm is not None else None

    # ---- the three collectives of the solver (overridable: tests stage them through the host)
    def halo_exchange(self, v_ext: torch.Tensor) -> None:
        """Fill v_ext[n_local:] with the peers' entries this rank's rows reference."""
        pl = self.plan
        if pl.n_send:
            self.ops.gather(pl.send_idx, v_ext, self.send_buf)
        recv = v_ext[self.n_local:self.n_local + pl.n_ghost]
        self._all_to_all(recv, self.send_buf[:pl.n_send], pl.recv_splits, pl.send_splits)

    def _all_to_all(self, recv, send, recv_splits, send_splits) -> None:
        if self.comm is not None:
            self.comm.all_to_all(recv, send, recv_splits, send_splits)
        else:
            dist.all_to_all_single(recv, send, recv_splits, send_splits, group=self.group)

    def gather_parts(self, dst: torch.Tensor, src: torch.Tensor) -> None:
        """dst[r*per:(r+1)*per] = rank r's chunk partials, in rank (= global chunk) order."""
        if self.comm is not None:
            self.comm.all_gather(dst, src)
        else:
            dist.all_gather_into_tensor(dst, src, group=self.group)

    def gather_parts_and_halo(self, dst_parts: torch.Tensor, src_parts: torch.Tensor, v_ext: torch.Tensor) -> None:
        """All-gather the chunk partials AND the halo of v in one step: pack v's entries the peers need, gather both
        buffers (one RCCL group), unpack the received slabs into v_ext's halo tail."""
        pl = self.plan
        if pl.n_send:
            self.ops.gather(pl.send_idx, v_ext, self.slab_loc)
        self._gather2(dst_parts, src_parts, self.slab_all, self.slab_loc)
        if pl.n_ghost:
            self.ops.gather(pl.ghost_src, self.slab_all, v_ext[self.n_local:self.n_local + pl.n_ghost])

    def _gather2(self, dst_a, src_a, dst_b, src_b) -> None:
        if self.comm is not None:
            self.comm.all_gather2(dst_a, src_a, dst_b, src_b)
        else:
            dist.all_gather_into_tensor(dst_a, src_a, group=self.group)
            dist.all_gather_into_tensor(dst_b, src_b, group=self.group)

    def agree_min(self, value: int) -> int:
        if self.part.world == 1:
            return value
        t = torch.tensor([value], dtype=torch.int64, device=self.ops.device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
        return int(t.item())


def native_loop_ok(prob: DistProblem) -> bool:
    """The C-driven loop (hipk_dist_cg_solve) needs the HIP kernels, a collective struct (direct RCCL, or whatever a
    subclass provides through `coll_struct()`), and rows on every rank."""
    import os
    part = prob.part
    return (isinstance(prob.ops, HipOps) and os.environ.get("HIPK_DIST_NATIVE", "1") != "0"
            and getattr(prob, "coll_struct", None) is not None and prob.coll_struct() is not None
            and part.per * (part.world - 1) < part.g and part.n_local > 0)


def _dist_cg_native(prob: DistProblem, x0_local, tol, atol, maxiter, check_every, solver: str = "cg", restart: int = 0,
                    solve_method: str = "batched"):
    """One call into libhipk.so runs the whole loop of this rank (csrc/hipk_dist.hip): the host enqueues fixed batches
    of iterations and reads the device stop word one batch late -- no Python between the kernels."""
    import os
    from . import _hipk
    L, part, pl = _hipk.lib(), prob.part, prob.plan
    dev = prob.ops.device
    n, n_ext = part.n_local, max(prob.n_ext, 1)
    peers = sum(1 for a, b in zip(pl.send_splits, pl.recv_splits) if a or b)
    mode = os.environ.get("HIPK_DIST_HALO", "p2p" if peers <= 4 else "allgather")
    coll = prob.coll_struct()
    if not coll.send or not coll.recv:
        mode = "allgather"          # a provider without send/recv (P2PComm): the halo rides in the gathered slabs
    plan = _hipk.DistPlan()
    plan.rank, plan.world = part.rank, part.world
    plan.n_local, plan.n_ext, plan.n_global = n, prob.n_ext, part.n_global
    plan.chunk_rows, plan.g_red, plan.per = part.ch, part.g, part.per
    plan.halo_mode = 1 if mode == "p2p" else 0
    plan.n_send, plan.n_ghost, plan.slab = pl.n_send, pl.n_ghost, pl.slab
    plan.send_idx_dev = pl.send_idx.data_ptr() if pl.n_send else None
    plan.ghost_src_dev = pl.ghost_src.data_ptr() if pl.n_ghost else None
    send_off_d, dest_off_d = pl.send_off.to(dev).contiguous(), pl.dest_off.to(dev).contiguous()   # fused exchanges (kept alive below)
    plan.send_off_dev, plan.dest_off_dev = send_off_d.data_ptr(), dest_off_d.data_ptr()
    sc = (ctypes.c_int32 * part.world)(*[int(v) for v in pl.send_splits])
    rc = (ctypes.c_int32 * part.world)(*[int(v) for v in pl.recv_splits])
    plan.send_counts, plan.recv_counts = sc, rc
    sf = (ctypes.c_int64 * part.world)(*[int(v) for v in pl.send_first])
    plan.send_first = sf
    x = prob.ops.zeros(n_ext)
    if x0_local is not None:
        x[:n] = x0_local
    if solver == "gmres":
        solve_fn = L.hipk_dist_gmres_solve
        wb = int(L.hipk_dist_gmres_work_bytes(ctypes.byref(plan), int(restart)))
    else:
        work_bytes_fn, solve_fn = {"cg": (L.hipk_dist_cg_work_bytes, L.hipk_dist_cg_solve),
                                   "bicgstab": (L.hipk_dist_bicgstab_work_bytes, L.hipk_dist_bicgstab_solve)}[solver]
        wb = int(work_bytes_fn(ctypes.byref(plan)))
    work = torch.empty(wb, dtype=torch.uint8, device=dev)
    prm = _hipk.Params()
    prm.tol, prm.atol = float(tol), float(atol)
    prm.maxiter = -1 if maxiter is None else int(maxiter)
    prm.check_every = int(check_every)
    if solver == "gmres":
        prm.restart = int(restart)
        prm.gmres_method = {"batched": 0, "incremental": 1}[solve_method]
        prm.gpu_tolerances = 1          # the reference's `device.type == 'cuda'` tolerance branch (TSL:737-740)
    st = _hipk.Stats()
    with torch.cuda.device(dev):
        rcode = solve_fn(prob.A["h"], ctypes.byref(plan), ctypes.byref(coll), prob.b.data_ptr(), x.data_ptr(),
                         work.data_ptr(), wb, ctypes.byref(prm), ctypes.byref(st),
                         torch.cuda.current_stream(dev).cuda_stream)
    _hipk._check(rcode, f"hipk_dist_{solver}_solve")
    if getattr(prob, "p2p", None) is not None and prob.p2p.failed():
        raise RuntimeError("hipk_p2p: a rank never published its part of an exchange (wait bound hit); results discarded")
    return x[:n], int(st.info), DistStats(int(st.iterations), int(st.matvecs), int(st.info), st.b_norm, st.residual_norm,
                                          st.x_norm, st.threshold)


def dist_bicgstab(prob: DistProblem, x0_local: Optional[torch.Tensor] = None, *, tol: float = 1e-5, atol: float = 0.0,
                  maxiter: Optional[int] = None, check_every: int = 16):
    """Row-partitioned BiCGStab (`hipk_dist_bicgstab_solve`): returns (x_local, info, DistStats); bit for bit the iterates of the
    single-device `bicgstab`.  Only the C-driven loop exists (HIP kernels + a collective struct: direct RCCL, the mailboxes, or a
    test's stand-ins) -- there is no backend-agnostic Python form of this solver."""
    if not native_loop_ok(prob):
        raise RuntimeError("dist_bicgstab needs the C-driven loop: HIP kernels, a collective provider and rows on every rank")
    return _dist_cg_native(prob, x0_local, tol, atol, maxiter, check_every, solver="bicgstab")


def dist_gmres(prob: DistProblem, x0_local: Optional[torch.Tensor] = None, *, tol: float = 1e-5, atol: float = 0.0,
               restart: int = 20, maxiter: Optional[int] = None, solve_method: str = "batched"):
    """Row-partitioned GMRES (`hipk_dist_gmres_solve`): returns (x_local, info, DistStats with iterations = restart cycles); bit
    for bit the iterates of the single-device `gmres` on a CUDA tensor.  C-driven loop only, restart <= 31."""
    if not native_loop_ok(prob):
        raise RuntimeError("dist_gmres needs the C-driven loop: HIP kernels, a collective provider and rows on every rank")
    if solve_method not in ("batched", "incremental"):
        raise ValueError(f"invalid solve_method {solve_method}, must be either 'batched' or 'incremental'")
    if not 1 <= int(restart) <= 31:
        raise ValueError("dist_gmres: restart must be in [1, 31]")
    return _dist_cg_native(prob, x0_local, tol, atol, maxiter, 0, solver="gmres", restart=restart, solve_method=solve_method)


def dist_cg(prob: DistProblem, x0_local: Optional[torch.Tensor] = None, *, tol: float = 1e-5, atol: float = 0.0,
            maxiter: Optional[int] = None, check_every: int = 32):
    """Row-partitioned CG; returns (x_local, info, DistStats). Same stopping rule, same `info` rule and,
    bit for bit, the same iterates as the single-device solve.  On GPUs with direct RCCL the whole loop runs in C
    (`_dist_cg_native`); the Python loop below is the backend-agnostic form (CPU test double, torch.distributed
    collectives, ranks without rows)."""
    if native_loop_ok(prob):
        return _dist_cg_native(prob, x0_local, tol, atol, maxiter, min(check_every, 16))
    ops, part, group = prob.ops, prob.part, prob.group
    n, ch, G, per, world = part.n_local, part.ch, part.g, part.per, part.world
    maxiter = 10 * part.n_global if maxiter is None else int(maxiter)
    n_ext = max(prob.n_ext, 1)
    x = ops.zeros(n_ext)                                    # x, p and r carry the halo tail
    if x0_local is not None:
        x[:n] = x0_local
    p, r = ops.zeros(n_ext), ops.zeros(n_ext)
    Ap = ops.zeros(max(n, 1))
    part_loc = ops.zeros(per)                               # this rank's chunk partials (zero padded)
    spare = ops.zeros(per)
    g_pAp, g_rr, g_bb = ops.zeros(world * per), ops.zeros(world * per), ops.zeros(world * per)
    scal = ops.scal_alloc()
    stop = ops.stop_word(scal)

    def gather_parts(dst):
        prob.gather_parts(dst, part_loc)

    # r0 = b - A x0, <r0,r0>; <b,b>   (TSL:815-826).  Halos of x and r0 are exchanged explicitly once;
    # inside the loop the halo of r rides with the <r,r> partials and every rank forms the halo entries of
    # p = r + beta p and x += alpha p itself (same operands, same bits as the owner).
    prob.halo_exchange(x)
    if n:
        ops.spmv(prob.A, x, r, MODE_RESID | MODE_DOT_YY, bsub=prob.b, part0=spare, part1=part_loc)
    gather_parts(g_rr)
    if n:
        ops.dot_parts(n, ch, prob.b, prob.b, part_loc)
    gather_parts(g_bb)
    prob.halo_exchange(r)
    if n:
        ops.cg_start(n, ch, G, scal, g_rr, g_bb, r, p, tol, atol, maxiter)
    if prob.plan.n_ghost:
        p[n:n + prob.plan.n_ghost] = r[n:n + prob.plan.n_ghost]          # p0 = r0 on the halo as well
    n_dir = prob.n_ext if n else 0
    it, stop_it = 0, None
    while it < maxiter:
        end = min(maxiter, it + check_every)
        while it < end:
            if n:
                ops.spmv(prob.A, p, Ap, MODE_DOT_W, w=p, part0=part_loc, part1=spare, stop=stop, it=it)
            gather_parts(g_pAp)
            if n:
                ops.cg_update(n, ch, G, scal, it, g_pAp, Ap, r, part_loc)
            prob.gather_parts_and_halo(g_rr, part_loc, r)
            if n:
                ops.cg_direction(n_dir, ch, G, scal, it, maxiter, g_pAp, g_rr, r, p, x)
            it += 1
        stop_it = _agree_stop(prob, scal)
        if stop_it <= it:
            break
    if stop_it is None:
        stop_it = _agree_stop(prob, scal)
    iterations = min(stop_it, it)
    # TSL:1007-1014: true residual and ||x|| decide info
    prob.halo_exchange(x)
    if n:
        ops.spmv(prob.A, x, Ap, MODE_RESID | MODE_DOT_YY, bsub=prob.b, part0=spare, part1=part_loc)
    gather_parts(g_rr)
    res2 = ops.reduce_parts(g_rr, G)
    if n:
        ops.dot_parts(n, ch, x, x, part_loc)
    gather_parts(g_pAp)
    xx = ops.reduce_parts(g_pAp, G)
    bs = ops.reduce_parts(g_bb, G)
    res2, xx, bs = float(res2.item()), float(xx.item()), float(bs.item())
    b_norm, res_norm, x_norm = max(bs, 0.0) ** 0.5, max(res2, 0.0) ** 0.5, (xx if xx == xx else float("nan"))
    x_norm = max(xx, 0.0) ** 0.5 if xx == xx else float("nan")
    thr = max(float(torch.tensor(tol, dtype=torch.float32).item()) * b_norm,
              float(torch.tensor(atol, dtype=torch.float32).item()))
    info = -1 if (x_norm != x_norm or res_norm > thr) else 0
    return x[:n], info, DistStats(iterations, iterations + 2, info, b_norm, res_norm, x_norm, thr)


def _agree_stop(prob, scal) -> int:
    """Every rank derives the same stop word from the same gathered partials; ranks without rows
    (more ranks than chunks) take it from the others."""
    s = prob.ops.read_scal(scal)["stop_it"] if prob.n_local else (1 << 62)
    return prob.agree_min(s)


class RowBlockCSR:
    """This rank's ROW BLOCK of a global square CSR matrix -- the operand that puts the row-partitioned solvers behind the
    reference's call surface (TSL:1019-1021, 1091-1093, 641-644; `SparseSolver.solve`, solver.py:256-379):

        dist.init_process_group("nccl", ...)                       # one process per GPU
        r0, r1 = RowBlockCSR.row_range(n_global)                    # the rows this rank owns
        A = RowBlockCSR(crow_local, col_global, values, n_global)   # rows r0 .. r1 of the global matrix, GLOBAL column ids
        x_local, info = cg(A, b[r0:r1], tol=1e-6)                   # or bicgstab / gmres / SparseSolver().solve(A, b_local, ...)

    Every rank makes the same call with its block; `b`, `x0` and the returned `x` are the rank's slices of the global vectors;
    `info` is the same on every rank.  Rows are split on reduction-chunk boundaries of the GLOBAL problem (`RowPartition`), so
    the iterates are bitwise those of the single-device solve.  The halo plan, the device matrix (coded SpMV form included) and
    the communicator are built on the first solve and reused by later ones (repeated solves with one matrix: the LDC caller).
    `M` (preconditioners), PyTrees, complex operands and autograd are not available on this operand (ValueError)."""
    _hipk_row_block = True

    def __init__(self, crow_local: torch.Tensor, col_global: torch.Tensor, values: torch.Tensor, n_global: int, *,
                 group=None, ops=None, problem_cls=None, force_ch: int = 0):
        if group is None and not dist.is_initialized():
            rank, world = 0, 1
        else:
            rank, world = dist.get_rank(group), dist.get_world_size(group)
        self.part = RowPartition(int(n_global), world, rank, force_ch)
        if crow_local.numel() != self.part.n_local + 1:
            raise ValueError(f"RowBlockCSR: rank {rank} of {world} owns rows [{self.part.row0}, {self.part.row1}) of the "
                             f"{n_global} x {n_global} system (RowBlockCSR.row_range), i.e. {self.part.n_local + 1} row pointers; "
                             f"got {crow_local.numel()}")
        if col_global.numel() != values.numel():
            raise ValueError("RowBlockCSR: col_global and values must have the same length")
        if values.dtype != torch.float64:
            raise ValueError("RowBlockCSR: fp64 values (the reference's working precision, TSL:979-980)")
        self.crow = (crow_local - crow_local[0]) if crow_local.numel() else crow_local
        self.col, self.val = col_global, values
        self.shape = (int(n_global), int(n_global))
        self.dtype, self.device = values.dtype, values.device
        self.group = group
        self._ops, self._problem_cls, self._prob = ops, problem_cls, None
        self.last_stats: Optional[DistStats] = None

    # ---- which rows a rank owns
    @staticmethod
    def row_range(n_global: int, group=None, rank: Optional[int] = None, world: Optional[int] = None, force_ch: int = 0):
        """(row_begin, row_end) of this rank's block (reduction-chunk aligned; a trailing rank may own no rows)."""
        if rank is None or world is None:
            rank, world = (dist.get_rank(group), dist.get_world_size(group)) if dist.is_initialized() else (0, 1)
        p = RowPartition(int(n_global), world, rank, force_ch)
        return p.row0, p.row1

    @classmethod
    def from_global_csr(cls, A: torch.Tensor, **kw):
        """Slice this rank's rows out of a replicated global CSR tensor (convenience; a large system is assembled per rank)."""
        if A.layout != torch.sparse_csr or A.shape[0] != A.shape[1]:
            raise ValueError("RowBlockCSR.from_global_csr needs a square sparse CSR tensor")
        r0, r1 = cls.row_range(A.shape[0], kw.get("group"), force_ch=kw.get("force_ch", 0))
        crow, col, val = A.crow_indices(), A.col_indices(), A.values()
        j0, j1 = int(crow[r0]), int(crow[r1])
        return cls((crow[r0:r1 + 1] - j0).clone(), col[j0:j1].clone(), val[j0:j1].clone(), A.shape[0], **kw)

    # ---- the solve behind cg / bicgstab / gmres
    def problem(self, b_local: torch.Tensor) -> "DistProblem":
        if self._prob is None:
            ops = self._ops
            if ops is None:
                if not self.val.is_cuda:
                    raise ValueError("RowBlockCSR: the row-partitioned solvers run the HIP kernels -- give CUDA/ROCm tensors "
                                     "(one process per GPU, backend 'nccl'); CPU tensors need an explicit `ops` backend (tests)")
                ops = HipOps(self.val.device)
            cls = self._problem_cls or DistProblem
            self._prob = cls(self.crow, self.col, self.val, b_local, self.part, ops, self.group)
        self._prob.b = b_local.contiguous()
        return self._prob

    def solve(self, method: str, b_local, x0_local=None, *, tol=1e-5, atol=0.0, maxiter=None, restart=20, solve_method="batched"):
        if not isinstance(b_local, torch.Tensor) or b_local.ndim != 1 or b_local.numel() != self.part.n_local:
            raise ValueError(f"RowBlockCSR: b must be this rank's slice of the right-hand side ({self.part.n_local} entries, rows "
                             f"[{self.part.row0}, {self.part.row1}))")
        if x0_local is not None and (not isinstance(x0_local, torch.Tensor) or x0_local.shape != b_local.shape):
            raise ValueError(f"arrays in x0 and b must have matching shapes: {getattr(x0_local, 'shape', None)} vs {b_local.shape}")
        if b_local.device != self.val.device:
            raise ValueError("RowBlockCSR: b lives on another device than the matrix block")
        prob = self.problem(b_local.detach().to(torch.float64))
        x0 = None if x0_local is None else x0_local.detach().to(torch.float64)
        if method == "cg":
            x, info, st = dist_cg(prob, x0, tol=tol, atol=atol, maxiter=maxiter)
        elif method == "bicgstab":
            x, info, st = dist_bicgstab(prob, x0, tol=tol, atol=atol, maxiter=maxiter)
        elif method == "gmres":
            x, info, st = dist_gmres(prob, x0, tol=tol, atol=atol, restart=restart, maxiter=maxiter, solve_method=solve_method)
        else:
            raise ValueError(f"Method '{method}' not available on a RowBlockCSR operand. Use: ['cg', 'bicgstab', 'gmres']")
        st.method = method
        self.last_stats = st
        return x.clone(), int(info), st

    def relative_residual(self) -> float:
        """||b - A x|| / ||b|| of the GLOBAL system for the last solve's x (its true-residual epilogue, TSL:1007-1014 / 766-773)."""
        st = self.last_stats
        return float("nan") if st is None else (st.residual_norm / st.b_norm if st.b_norm > 0 else float("inf"))

    @classmethod
    def poisson5(cls, nx: int, ny: int, *, device=None, **kw):
        """bench.py workload: this rank's rows of the 5-point Poisson matrix on an nx x ny grid (row k = i * ny + j)."""
        from .utils.matrix_utils import stencil5_csr_components
        r0, r1 = cls.row_range(nx * ny, kw.get("group"), force_ch=kw.get("force_ch", 0))
        crow, col, val = stencil5_csr_components(nx, ny, 4.0, -1.0, -1.0, -1.0, -1.0, row_begin=r0, row_end=r1, device=device)
        return cls(crow, col, val, nx * ny, **kw)


class DistPoissonProblem(DistProblem):
    """bench.py workloads: 5-point Poisson on an nx x ny grid (row k = i*ny + j), b = ones, rows split by RowPartition
    (contiguous blocks on reduction-chunk boundaries of the GLOBAL problem -- a rank's block need not end on a grid line).
      weak scaling   nx = nx_per_rank * world  (every rank about nx_per_rank grid lines: per-GPU work fixed)
      strong scaling nx = nx_global            (BASELINE config 5: 8000 x 8000, N = 64 M, over 1/2/4/8 ranks)"""

    def __init__(self, nx_per_rank: int = 0, ny: int = 0, rank: int = 0, world: int = 1, device=None, group=None,
                 force_ch: int = 0, nx_global: int = 0):
        from .utils.matrix_utils import stencil5_csr_components
        assert (nx_per_rank > 0) != (nx_global > 0), "give nx_per_rank (weak) or nx_global (strong)"
        ops = HipOps(device)
        nx = nx_global if nx_global > 0 else nx_per_rank * world
        part = RowPartition(nx * ny, world, rank, force_ch)
        crow, col, val = stencil5_csr_components(nx, ny, 4.0, -1.0, -1.0, -1.0, -1.0, row_begin=part.row0,
                                                 row_end=part.row1, device=device)
        b = torch.ones(part.n_local, dtype=torch.float64, device=device)
        super().__init__(crow, col, val, b, part, ops, group)
