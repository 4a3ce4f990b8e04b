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
from pytorch_sparse_solver.distributed import DistProblem, RowPartition, dist_cg  # noqa: E402
from pytorch_sparse_solver.utils.matrix_utils import create_poisson_2d_csr  # noqa: E402


def build_global(kind, nx, ny):
    A = create_poisson_2d_csr(nx, ny)
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


def main():
    kind, nx, ny, tol, maxiter, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
    use_hip = len(sys.argv) > 7 and sys.argv[7] == "hip"
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
        prob = HostStagedProblem.__new__(HostStagedProblem)
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
    else:
        prob = DistProblem(lc, lcol, lval, lb, part, OracleOps())
    x_loc, info, st = dist_cg(prob, tol=tol, maxiter=None if maxiter < 0 else maxiter, check_every=7)
    pieces = [None] * world
    dist.all_gather_object(pieces, (part.row0, x_loc.cpu().numpy().copy(), info, st.iterations, st.residual_norm))
    if rank == 0:
        x = np.concatenate([p[1] for p in sorted(pieces, key=lambda q: q[0])])
        ref = O.cg(crow.numpy(), col.numpy(), val.numpy(), b.numpy(), tol=tol, maxiter=None if maxiter < 0 else maxiter)
        res = {"bitwise_equal": bool(np.array_equal(x, ref.x)), "info": [p[2] for p in pieces], "ref_info": ref.info,
               "iterations": [p[3] for p in pieces], "ref_iterations": ref.iterations,
               "residual_norm": [p[4] for p in pieces], "ref_residual_norm": ref.residual_norm,
               "n_local": [int(p[1].size) for p in sorted(pieces, key=lambda q: q[0])],
               "n_ghost": prob.plan.n_ghost, "chunk": part.ch, "chunks": part.g}
        with open(out, "w") as f:
            json.dump(res, f)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
