#!/usr/bin/env python3
"""Generate tests/golden/* by RUNNING THE REFERENCE (imported from /root/reference/src).

Runs only in the build container (the reference never travels to the GPU box).  The
fixtures are data: input arrays (CSR, b, x0) and the reference's outputs
(x, info, number of operator applications, true relative residual) -- no reference
source.  Usage:  PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py

Problems:
  poisson_nx{8,16,32,64}   5-pt Poisson from the reference's own builder, b = ones
  convdiff_nx{16,32,64}    synthetic nonsymmetric 5-pt convection-diffusion, b = A randn
  ldc_nx{8,16,32}          lid-driven-cavity pressure matrix + RHS of FVM steps 0..2,
                           produced by the reference's BaseLDCSolver
  spd_n{50,100}            dense SPD (test_module_a.py:39-42 recipe) stored as CSR
  tridiag_n100             test_module_a.py:93-124 recipe
Each solver run records x and scalars; operator applications are counted by passing the
reference a counting callable (its tensor path computes the same torch.matmul).
"""
import json
import os
import sys
import warnings

import numpy as np
import torch

warnings.filterwarnings("ignore")
REF = "/root/reference"
sys.path.insert(0, os.path.join(REF, "src"))
sys.path.insert(0, os.path.join(REF, "FVM_example", "LDC_by_torchsp"))
sys.dont_write_bytecode = True

from pytorch_sparse_solver.module_a import bicgstab, cg, gmres  # noqa: E402
from pytorch_sparse_solver.solver import SparseSolver  # noqa: E402
from pytorch_sparse_solver.utils.matrix_utils import create_poisson_2d_sparse_coo  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
OUT = os.path.normpath(OUT)
SOLVERS = {"cg": cg, "bicgstab": bicgstab, "gmres": gmres}


class Counting:
    def __init__(self, A):
        self.A = A
        self.count = 0

    def __call__(self, v):
        self.count += 1
        return torch.matmul(self.A, v)


def csr_parts(A_csr):
    return (A_csr.crow_indices().numpy().astype(np.int32), A_csr.col_indices().numpy().astype(np.int32),
            A_csr.values().numpy().astype(np.float64))


def convdiff_coo(nx, ny, gamma=0.5, delta=0.25):
    """Synthetic config-3 matrix: same ordering as the Poisson builder."""
    rows, cols, vals = [], [], []
    for i in range(nx):
        for j in range(ny):
            k = i * ny + j
            rows.append(k); cols.append(k); vals.append(4.0)
            if i > 0:
                rows.append(k); cols.append((i - 1) * ny + j); vals.append(-1.0 - gamma)
            if i < nx - 1:
                rows.append(k); cols.append((i + 1) * ny + j); vals.append(-1.0 + gamma)
            if j > 0:
                rows.append(k); cols.append(k - 1); vals.append(-1.0 - delta)
            if j < ny - 1:
                rows.append(k); cols.append(k + 1); vals.append(-1.0 + delta)
    idx = torch.tensor([rows, cols], dtype=torch.long)
    return torch.sparse_coo_tensor(idx, torch.tensor(vals, dtype=torch.float64), (nx * ny, nx * ny)).coalesce()


def run_case(index, arrays, name, A_csr, b, runs, x0=None):
    """runs: list of (tag, solver_name, kwargs)."""
    for tag, sname, kw in runs:
        op = Counting(A_csr)
        kwargs = dict(kw)
        if x0 is not None:
            kwargs["x0"] = x0
        x, info = SOLVERS[sname](op, b, **kwargs)
        # the tensor path must agree with the counted callable path
        x_t, info_t = SOLVERS[sname](A_csr, b, **kwargs)
        assert info_t == info and torch.allclose(x_t, x, rtol=1e-9, atol=1e-12), (name, tag)
        res = torch.norm(b - A_csr @ x).item()
        arrays[f"{tag}_x"] = x.numpy()
        index.append({
            "case": name, "tag": tag, "solver": sname,
            "kwargs": {k: v for k, v in kw.items()},
            "has_x0": x0 is not None,
            "info": int(info), "matvecs": int(op.count),
            "residual_norm": res, "b_norm": torch.norm(b).item(),
            "x_norm": torch.norm(x).item(),
        })
        print(f"  {name:16s} {tag:22s} info={info:2d} matvecs={op.count:5d} relres={res / max(torch.norm(b).item(), 1e-300):.3e}")


def save_case(name, A_csr, b, arrays, x0=None):
    crow, col, val = csr_parts(A_csr)
    extra = {} if x0 is None else {"x0": x0.numpy()}
    np.savez_compressed(os.path.join(OUT, name + ".npz"), crow=crow, col=col, val=val, b=b.numpy(),
                        n=np.int64(A_csr.shape[0]), **extra, **arrays)


def std_runs(tol, maxiter=None, restart=30, with_cg=True):
    base = {"tol": tol}
    if maxiter is not None:
        base["maxiter"] = maxiter
    runs = []
    if with_cg:
        runs.append(("cg", "cg", dict(base)))
    runs.append(("bicgstab", "bicgstab", dict(base)))
    runs.append(("gmres_batched", "gmres", dict(base, restart=restart, solve_method="batched")))
    runs.append(("gmres_incremental", "gmres", dict(base, restart=restart, solve_method="incremental")))
    return runs


def main():
    os.makedirs(OUT, exist_ok=True)
    index = []
    torch.set_num_threads(4)

    # ---- Poisson (reference builder: utils/matrix_utils.py:193-257)
    for nx in (8, 16, 32, 64):
        A = create_poisson_2d_sparse_coo(nx, nx).to_sparse_csr()
        b = torch.ones(nx * nx, dtype=torch.float64)
        arrays = {}
        name = f"poisson_nx{nx}"
        run_case(index, arrays, name, A, b, std_runs(1e-6))
        save_case(name, A, b, arrays)
    # ragged grid + warm start + atol + maxiter cut-off
    A = create_poisson_2d_sparse_coo(17, 13).to_sparse_csr()
    g = torch.Generator().manual_seed(7)
    b = torch.randn(17 * 13, dtype=torch.float64, generator=g)
    x0 = torch.randn(17 * 13, dtype=torch.float64, generator=g)
    arrays = {}
    runs = [("cg_x0", "cg", {"tol": 1e-8}), ("cg_maxiter5", "cg", {"tol": 1e-12, "maxiter": 5}),
            ("cg_atol", "cg", {"tol": 0.0, "atol": 1e-3}),
            ("bicgstab_x0", "bicgstab", {"tol": 1e-8}),
            ("bicgstab_maxiter3", "bicgstab", {"tol": 1e-12, "maxiter": 3}),
            ("gmres_r5", "gmres", {"tol": 1e-8, "restart": 5, "solve_method": "batched"}),
            ("gmres_r5_inc", "gmres", {"tol": 1e-8, "restart": 5, "solve_method": "incremental"}),
            ("gmres_default", "gmres", {}),
            ("gmres_maxiter2", "gmres", {"tol": 1e-12, "restart": 4, "maxiter": 2})]
    run_case(index, arrays, "poisson_17x13", A, b, runs, x0=x0)
    save_case("poisson_17x13", A, b, arrays, x0=x0)

    # ---- convection-diffusion (synthetic; RHS = A randn as test_module_a.py:144-145 does)
    for nx in (16, 32, 64):
        A = convdiff_coo(nx, nx).to_sparse_csr()
        g = torch.Generator().manual_seed(0)
        b = A @ torch.randn(nx * nx, dtype=torch.float64, generator=g)
        arrays = {}
        name = f"convdiff_nx{nx}"
        run_case(index, arrays, name, A, b, std_runs(1e-6, with_cg=False))
        save_case(name, A, b, arrays)

    # ---- LDC pressure systems (FVM_example/LDC_by_torchsp/ldc_solver_common.py:90-135, 185-201)
    from ldc_solver_common import BaseLDCSolver

    class Recorder(BaseLDCSolver):
        def __init__(self, *a, **k):
            self.rhs = []
            super().__init__(*a, solver_label="golden", **k)

        def _solve_linear_system(self, prhs):
            self.rhs.append(prhs.clone())
            # exactly what ldc_solver_module_a.py:21 calls
            return gmres(self.A_dense, prhs, tol=1e-10, maxiter=1000, restart=30)

    for nx in (8, 16, 32):
        import contextlib
        import io
        with contextlib.redirect_stdout(io.StringIO()):
            s = Recorder(nx=nx, Re=100.0, method="gmres", device="cpu")
            for _ in range(3):
                s.step()
        A = s.A_csr
        for step, prhs in enumerate(s.rhs):
            arrays = {}
            name = f"ldc_nx{nx}_step{step}"
            runs = [("bicgstab", "bicgstab", {"tol": 1e-10, "maxiter": 1000}),
                    ("gmres_batched", "gmres", {"tol": 1e-10, "maxiter": 1000, "restart": 30,
                                                "solve_method": "batched"}),
                    ("gmres_incremental", "gmres", {"tol": 1e-10, "maxiter": 1000, "restart": 30,
                                                    "solve_method": "incremental"})]
            run_case(index, arrays, name, A, prhs, runs)
            save_case(name, A, prhs, arrays)

    # ---- dense SPD as CSR (test_module_a.py:39-42) and the tridiagonal recipe (:93-124)
    for n in (50, 100):
        g = torch.Generator().manual_seed(n)
        G = torch.randn(n, n, dtype=torch.float64, generator=g)
        Ad = G @ G.T + n * torch.eye(n, dtype=torch.float64)
        A = Ad.to_sparse_csr()
        b = Ad @ torch.randn(n, dtype=torch.float64, generator=g)
        arrays = {}
        name = f"spd_n{n}"
        run_case(index, arrays, name, A, b, std_runs(1e-8, maxiter=500))
        save_case(name, A, b, arrays)
    n = 100
    Ad = 2 * torch.eye(n, dtype=torch.float64) - torch.diag(torch.ones(n - 1, dtype=torch.float64), 1) \
        - torch.diag(torch.ones(n - 1, dtype=torch.float64), -1)
    A = Ad.to_sparse_csr()
    g = torch.Generator().manual_seed(1)
    b = Ad @ torch.randn(n, dtype=torch.float64, generator=g)
    arrays = {}
    run_case(index, arrays, "tridiag_n100", A, b, std_runs(1e-10, maxiter=1000))
    save_case("tridiag_n100", A, b, arrays)

    # ---- dispatcher record (solver.py:320-379) on one case
    A = create_poisson_2d_sparse_coo(16, 16).to_sparse_csr()
    b = torch.ones(256, dtype=torch.float64)
    disp = []
    for method, kw in (("cg", {}), ("bicgstab", {}), ("gmres", {"restart": 30})):
        x, res = SparseSolver().solve(A, b, method=method, backend="module_a", tol=1e-6, **kw)
        disp.append({"method": method, "kwargs": kw, "converged": bool(res.converged), "residual": res.residual,
                     "iterations": res.iterations, "backend": res.backend, "x_norm": torch.norm(x).item()})
    meta = {"generator": "oracle/gen_golden.py", "torch": torch.__version__,
            "reference": "Litianyu141/Pytorch-Sparse-Linalg-torch-amgx.cg.bicg.gmres @ /root/reference",
            "runs": index, "dispatcher_poisson_nx16": disp}
    with open(os.path.join(OUT, "index.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print(f"wrote {len(index)} runs to {OUT}")


if __name__ == "__main__":
    main()
