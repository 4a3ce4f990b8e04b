"""Worker of tests/test_distributed_api.py: one rank that reaches the row-partitioned solvers ONLY through the reference's call
surface -- `SparseSolver().solve(A, b, method=..., backend='module_a')` and `module_a.cg / bicgstab / gmres` -- with a
`RowBlockCSR` operand.  cpu: gloo + the CPU ops double; hip: several ranks share cuda:0, the C-driven loops
run with host-staged stand-ins for the collectives (tests/_dist_worker.py)."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd"), os.path.join(ROOT, "tests")]

from _dist_worker import HostStagedNative, build_global  # noqa: E402
from dist_cpu_ops import OracleOps  # noqa: E402
from oracle import oracle as O  # noqa: E402
import pytorch_sparse_solver as pss  # noqa: E402
from pytorch_sparse_solver import SparseSolver  # noqa: E402
from pytorch_sparse_solver import module_a  # noqa: E402


class HostStagedNativeFromBlock(HostStagedNative):
    """DistProblem constructor for the shared-GPU rehearsal: the halo plan's set-up collectives run on CPU tensors (gloo), the
    matrix block and the vectors live on cuda:0."""

    def __init__(self, crow, col_global, val, b_local, part, ops, group=None):
        from pytorch_sparse_solver.distributed import HaloPlan
        dev = ops.device
        plan = HaloPlan(col_global.cpu(), part)
        for name in ("col_local", "send_idx", "ghost_src"):
            setattr(plan, name, getattr(plan, name).to(dev))
        self.part, self.ops, self.group, self.plan = part, ops, None, plan
        self.n_local, self.n_ext, self.nnz_local = part.n_local, part.n_local + plan.n_ghost, int(val.numel())
        self.b = b_local.to(dev)
        self.A = ops.make_matrix(crow.to(dev), plan.col_local, val.to(dev), part.n_local, max(self.n_ext, 1), part.ch) \
            if part.n_local else None
        self.spmv_bytes = 0
        self.send_buf = ops.empty(max(plan.n_send, 1))
        self.slab_loc, self.slab_all = ops.zeros(plan.slab), ops.zeros(plan.slab * part.world)
        self.comm, self.p2p, self.comm_kind = None, None, "host-staged (test)"


def main():
    kind, nx, ny, tol, maxiter, out, mode, solver, entry = sys.argv[1:10]
    nx, ny, tol, maxiter = int(nx), int(ny), float(tol), int(maxiter)
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    A, b = build_global(kind, nx, ny)
    n = nx * ny
    r0, r1 = pss.RowBlockCSR.row_range(n)
    if mode == "hip":
        from pytorch_sparse_solver.distributed import HipOps
        dev = torch.device("cuda", 0)
        Arb = pss.RowBlockCSR.from_global_csr(A.to(dev), ops=HipOps(dev), problem_cls=HostStagedNativeFromBlock)
        b_loc = b[r0:r1].to(dev)
    else:
        Arb = pss.RowBlockCSR.from_global_csr(A, ops=OracleOps())
        b_loc = b[r0:r1].clone()
    kw = {"tol": tol}
    if maxiter >= 0:
        kw["maxiter"] = maxiter
    method = "gmres" if solver.startswith("gmres") else solver
    if method == "gmres":
        kw.update(restart=12, solve_method="incremental" if solver.endswith("incremental") else "batched")
    record = None
    if entry == "solver":          # SparseSolver.solve(..., backend='module_a') (solver.py:256-379)
        x_loc, res = SparseSolver().solve(Arb, b_loc, method=method, backend="module_a", **kw)
        info = 0 if res.converged else -1
        record = {"converged": bool(res.converged), "residual": res.residual, "backend": res.backend, "method": res.method,
                  "iterations": res.iterations}
    else:                          # module_a.cg / bicgstab / gmres (TSL:1019, 1091, 641)
        x_loc, info = getattr(module_a, method)(Arb, b_loc, **kw)
    st = module_a.get_last_stats()
    # a second solve on the same operand (cached plan / device matrix / communicator), warm-started from the first result
    x2, info2 = getattr(module_a, method)(Arb, b_loc, x0=x_loc, **kw)
    pieces = [None] * world
    dist.all_gather_object(pieces, (r0, x_loc.cpu().numpy().copy(), int(info), st.iterations, st.residual_norm, record,
                                    x2.cpu().numpy().copy(), int(info2)))
    if rank == 0:
        pieces.sort(key=lambda q: q[0])
        x = np.concatenate([p[1] for p in pieces])
        x2g = np.concatenate([p[6] for p in pieces])
        crow, col, val = A.crow_indices().numpy(), A.col_indices().numpy(), A.values().numpy()
        okw = dict(tol=tol, maxiter=None if maxiter < 0 else maxiter)
        if method == "gmres":
            ref = O.gmres(crow, col, val, b.numpy(), restart=12, solve_method=kw["solve_method"], gpu_tolerances=True, **okw)
            ref2 = O.gmres(crow, col, val, b.numpy(), x0=ref.x, restart=12, solve_method=kw["solve_method"], gpu_tolerances=True, **okw)
        else:
            ref = getattr(O, method)(crow, col, val, b.numpy(), **okw)
            ref2 = getattr(O, method)(crow, col, val, b.numpy(), x0=ref.x, **okw)
        res = {"bitwise_equal": bool(np.array_equal(x, ref.x)), "info": [p[2] for p in pieces], "ref_info": ref.info,
               "iterations": [p[3] for p in pieces], "ref_iterations": ref.iterations,
               "residual_norm": [p[4] for p in pieces], "ref_residual_norm": ref.residual_norm, "ref_b_norm": ref.b_norm,
               "records": [p[5] for p in pieces], "n_local": [int(p[1].size) for p in pieces],
               "second_bitwise_equal": bool(np.array_equal(x2g, ref2.x)), "second_info": [p[7] for p in pieces],
               "ref2_info": ref2.info}
        with open(out, "w") as f:
            json.dump(res, f)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
