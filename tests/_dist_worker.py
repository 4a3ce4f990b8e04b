"""Worker of tests/test_distributed_gloo.py: one gloo rank of the row-partitioned CG on CPU."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd"), os.path.join(ROOT, "tests")]

from dist_cpu_ops import OracleOps  # noqa: E402
from oracle import oracle as O  # noqa: E402
from pytorch_sparse_solver.distributed import DistProblem, RowPartition, dist_bicgstab, dist_cg, dist_gmres  # noqa: E402
from pytorch_sparse_solver.utils.matrix_utils import create_convdiff_2d_csr, create_poisson_2d_csr  # noqa: E402


def build_global(kind, nx, ny):
    A = create_convdiff_2d_csr(nx, ny) if kind == "convdiff" else create_poisson_2d_csr(nx, ny)
    n = nx * ny
    if kind == "random_spd":
        # add symmetric long-range couplings (ghosts from several owners), keep diagonal dominance
        g = torch.Generator().manual_seed(5)
        m = n // 2
        i = torch.randint(0, n, (m,), generator=g)
        j = torch.randint(0, n, (m,), generator=g)
        keep = i != j
        i, j = i[keep], j[keep]
        v = -0.1 * torch.rand(i.numel(), dtype=torch.float64, generator=g)
        idx = torch.cat([torch.stack([i, j]), torch.stack([j, i]), torch.stack([torch.arange(n), torch.arange(n)])], dim=1)
        val = torch.cat([v, v, torch.full((n,), 2.0, dtype=torch.float64)])
        A = (A.to_sparse_coo() + torch.sparse_coo_tensor(idx, val, (n, n))).coalesce().to_sparse_csr()
    g = torch.Generator().manual_seed(11)
    b = torch.randn(n, dtype=torch.float64, generator=g)
    return A, b


class HostStagedProblem(DistProblem):
    """HIP kernels on this process's GPU, collectives staged through the host (gloo): lets several ranks
    share ONE GPU so the step API is exercised under a real multi-rank partition on the 1-GPU box."""

    def _all_to_all(self, recv, send, recv_splits, send_splits):
        r = torch.empty(recv.numel(), dtype=recv.dtype)
        dist.all_to_all_single(r, send.cpu(), recv_splits, send_splits)
        recv.copy_(r)

    def gather_parts(self, dst, src):
        d = torch.empty(dst.numel(), dtype=dst.dtype)
        dist.all_gather_into_tensor(d, src.cpu())
        dst.copy_(d)

    def _gather2(self, dst_a, src_a, dst_b, src_b):
        self.gather_parts(dst_a, src_a)
        self.gather_parts(dst_b, src_b)

    def agree_min(self, value):
        t = torch.tensor([value], dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return int(t.item())


class HostStagedNative(HostStagedProblem):
    """The C-DRIVEN loop (hipk_dist_cg_solve) under a real multi-rank partition on ONE GPU: the hipk_rccl entry points are
    Python callbacks that stage the collectives through the host over gloo (RCCL refuses several ranks on one device).
    Calls between group_start and group_end are deferred to group_end, like RCCL does."""

    def coll_struct(self):
        import ctypes
        from pytorch_sparse_solver import _hipk
        if getattr(self, "_coll", None) is not None:
            return self._coll
        hip = ctypes.CDLL("libamdhip64.so")
        hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        hip.hipStreamSynchronize.argtypes = [ctypes.c_void_p]
        D2H, H2D = 2, 1
        pending, depth = [], [0]

        def to_host(ptr, count):
            a = np.empty(count, dtype=np.float64)
            assert hip.hipMemcpy(a.ctypes.data, ptr, count * 8, D2H) == 0
            return torch.from_numpy(a)

        def to_dev(ptr, t):
            a = np.ascontiguousarray(t.numpy())
            assert hip.hipMemcpy(ptr, a.ctypes.data, a.size * 8, H2D) == 0

        def run(ops, stream):
            hip.hipStreamSynchronize(stream)
            p2p, keep = [], []
            for op in ops:
                if op[0] == "ag":
                    _, send, recv, count = op
                    out = torch.empty(count * dist.get_world_size(), dtype=torch.float64)
                    dist.all_gather_into_tensor(out, to_host(send, count))
                    to_dev(recv, out)
                elif op[0] == "send":
                    p2p.append(dist.P2POp(dist.isend, to_host(op[1], op[2]), op[3]))
                else:
                    buf = torch.empty(op[2], dtype=torch.float64)
                    keep.append((op[1], buf))
                    p2p.append(dist.P2POp(dist.irecv, buf, op[3]))
            if p2p:
                for w in dist.batch_isend_irecv(p2p):
                    w.wait()
            for ptr, buf in keep:
                to_dev(ptr, buf)

        def issue(op, stream):
            try:
                if depth[0] > 0:
                    pending.append((op, stream))
                else:
                    run([op], stream)
                return 0
            except BaseException as e:   # never unwind through the C frames
                print("collective callback failed:", repr(e), flush=True)
                return 1

        def group_start():
            depth[0] += 1
            return 0

        def group_end():
            depth[0] -= 1
            if depth[0] == 0 and pending:
                ops, stream = [o for o, _ in pending], pending[0][1]
                pending.clear()
                try:
                    run(ops, stream)
                except BaseException as e:
                    print("collective callback failed:", repr(e), flush=True)
                    return 1
            return 0

        def all_gather(send, recv, count, dtype, comm, stream):
            assert dtype == 8
            return issue(("ag", send, recv, count), stream)

        def send(buf, count, dtype, peer, comm, stream):
            return issue(("send", buf, count, peer), stream)

        def recv(buf, count, dtype, peer, comm, stream):
            return issue(("recv", buf, count, peer), stream)

        self._cbs = (_hipk.COLL_GROUP_FN(group_start), _hipk.COLL_GROUP_FN(group_end), _hipk.COLL_ALLGATHER_FN(all_gather),
                     _hipk.COLL_SENDRECV_FN(send), _hipk.COLL_SENDRECV_FN(recv))
        addr = lambda f: ctypes.cast(f, ctypes.c_void_p).value   # noqa: E731
        self._coll = _hipk.Rccl(*[addr(f) for f in self._cbs], None)
        return self._coll


class MailboxNative(HostStagedProblem):
    """The C-driven loop with the product's device-mailbox exchange (P2PComm, csrc/hipk_p2p.hip): several ranks on ONE
    GPU map each other's mailboxes through HIP IPC -- no host staging anywhere in the iteration."""

    def coll_struct(self):
        return self.p2p.coll_struct()


def main():
    kind, nx, ny, tol, maxiter, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
    use_hip = len(sys.argv) > 7 and sys.argv[7] in ("hip", "native", "native_ag", "native_p2p", "native_side", "native_fused")
    mailbox = len(sys.argv) > 7 and sys.argv[7] in ("native_p2p", "native_fused")
    fused = len(sys.argv) > 7 and sys.argv[7] == "native_fused"   # exchanges made by the CG kernels themselves (csrc/hipk_fx.h)
    native = len(sys.argv) > 7 and sys.argv[7].startswith("native")
    if len(sys.argv) > 7 and sys.argv[7] == "native_side":
        os.environ["HIPK_DIST_OVERLAP"] = "1"      # x += alpha p on a side stream beside the second collective
    if len(sys.argv) > 7 and sys.argv[7] == "native_ag":
        os.environ["HIPK_DIST_HALO"] = "allgather"
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    A, b = build_global(kind, nx, ny)
    n = nx * ny
    part = RowPartition(n, world, rank)
    crow, col, val = A.crow_indices(), A.col_indices(), A.values()
    j0, j1 = int(crow[part.row0]), int(crow[part.row1])
    lc, lcol, lval, lb = (crow[part.row0:part.row1 + 1] - j0).clone(), col[j0:j1].clone(), val[j0:j1].clone(), \
        b[part.row0:part.row1].clone()
    if use_hip:
        from pytorch_sparse_solver.distributed import HaloPlan, HipOps
        plan = HaloPlan(lcol, part)                       # plan collectives on CPU tensors (gloo)
        dev = torch.device("cuda", 0)
        cls = MailboxNative if mailbox else HostStagedNative if native else HostStagedProblem
        prob = cls.__new__(cls)
        ops = HipOps(dev)
        for name in ("col_local", "send_idx", "ghost_src"):
            setattr(plan, name, getattr(plan, name).to(dev))
        prob.part, prob.ops, prob.group, prob.plan = part, ops, None, plan
        prob.n_local, prob.n_ext, prob.nnz_local = part.n_local, part.n_local + plan.n_ghost, int(lval.numel())
        prob.b = lb.to(dev)
        prob.A = ops.make_matrix(lc.to(dev), plan.col_local, lval.to(dev), part.n_local, max(prob.n_ext, 1), part.ch) \
            if part.n_local else None
        prob.spmv_bytes = 0
        prob.send_buf = ops.empty(max(plan.n_send, 1))
        prob.slab_loc, prob.slab_all = ops.zeros(plan.slab), ops.zeros(plan.slab * world)
        prob.comm = None
        if mailbox:
            from pytorch_sparse_solver.distributed import P2PComm
            prob.p2p = P2PComm(rank, world, dev, max(part.per, plan.slab), fx_per=part.per if fused else 0,
                               fx_ghost_cap=plan.ghost_cap if fused else 0)
    else:
        prob = DistProblem(lc, lcol, lval, lb, part, OracleOps())
    if native:
        from pytorch_sparse_solver.distributed import native_loop_ok
        assert native_loop_ok(prob), "the C-driven loop was expected to run"
    solver = sys.argv[8] if len(sys.argv) > 8 else "cg"
    if solver.startswith("gmres"):     # "gmres" / "gmres_incremental": restart 12, cycles capped by maxiter
        method = "incremental" if solver.endswith("incremental") else "batched"
        x_loc, info, st = dist_gmres(prob, tol=tol, restart=12, maxiter=None if maxiter < 0 else maxiter, solve_method=method)
    else:
        solve = dist_bicgstab if solver == "bicgstab" else dist_cg
        x_loc, info, st = solve(prob, tol=tol, maxiter=None if maxiter < 0 else maxiter, check_every=7)
    kname = ""
    if use_hip:
        from pytorch_sparse_solver import _hipk
        kname = _hipk.CsrHandle.last_spmv_kernel()      # the SpMV kernel this rank's last launch selected
    pieces = [None] * world
    dist.all_gather_object(pieces, (part.row0, x_loc.cpu().numpy().copy(), info, st.iterations, st.residual_norm, kname))
    if rank == 0:
        x = np.concatenate([p[1] for p in sorted(pieces, key=lambda q: q[0])])
        if solver.startswith("gmres"):
            ref = O.gmres(crow.numpy(), col.numpy(), val.numpy(), b.numpy(), tol=tol, restart=12, maxiter=None if maxiter < 0 else maxiter,
                          solve_method="incremental" if solver.endswith("incremental") else "batched", gpu_tolerances=True)
        else:
            ref = (O.bicgstab if solver == "bicgstab" else O.cg)(crow.numpy(), col.numpy(), val.numpy(), b.numpy(), tol=tol,
                                                                  maxiter=None if maxiter < 0 else maxiter)
        res = {"bitwise_equal": bool(np.array_equal(x, ref.x)), "info": [p[2] for p in pieces], "ref_info": ref.info,
               "iterations": [p[3] for p in pieces], "ref_iterations": ref.iterations,
               "residual_norm": [p[4] for p in pieces], "ref_residual_norm": ref.residual_norm,
               "n_local": [int(p[1].size) for p in sorted(pieces, key=lambda q: q[0])],
               "n_ghost": prob.plan.n_ghost, "chunk": part.ch, "chunks": part.g,
               "spmv_kernel": [p[5] for p in sorted(pieces, key=lambda q: q[0])]}
        with open(out, "w") as f:
            json.dump(res, f)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
