#!/usr/bin/env python3
"""Full-size reference runs (BASELINE configs 2 and 3, N = 4,000,000) -> tests/golden/big_index.json.

Runs THE REFERENCE (imported from /root/reference/src) on CPU; only scalars and 16 sampled entries of x are
stored ("for the big configs only scalars").  Takes a few minutes; build container only.
Usage: PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_big.py
"""
import json
import os
import sys
import time
import warnings

import torch

warnings.filterwarnings("ignore")
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference/src")
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from pytorch_sparse_solver.module_a import bicgstab, cg  # noqa: E402  (the REFERENCE package: first on sys.path)
import pytorch_sparse_solver  # noqa: E402

assert pytorch_sparse_solver.__file__.startswith("/root/reference"), pytorch_sparse_solver.__file__

# vectorised builders of this repo (bit-identical to the reference's loop builders, tests/test_matrix_utils)
import importlib.util  # noqa: E402
spec = importlib.util.spec_from_file_location(
    "mu", os.path.join(ROOT, "pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd", "pytorch_sparse_solver", "utils",
                       "matrix_utils.py"))
mu = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mu)


class Counting:
    def __init__(self, A):
        self.A, self.count = A, 0

    def __call__(self, v):
        self.count += 1
        return torch.matmul(self.A, v)


def record(name, solver, A, b, **kw):
    op = Counting(A)
    t0 = time.time()
    x, info = solver(op, b, **kw)
    dt = time.time() - t0
    n = b.numel()
    idx = [(i * 2654435761) % n for i in range(16)]
    res = torch.norm(b - A @ x).item()
    out = {"case": name, "n": n, "kwargs": kw, "info": int(info), "matvecs": op.count, "relres": res / torch.norm(b).item(),
           "x_norm": torch.norm(x).item(), "sample_idx": idx, "sample_x": [x[i].item() for i in idx], "seconds": dt}
    print(json.dumps(out), flush=True)
    return out


def main():
    torch.set_num_threads(8)
    runs = []
    for nx in (1000, 2000):
        A = mu.create_poisson_2d_csr(nx, nx)
        b = torch.ones(nx * nx, dtype=torch.float64)
        runs.append(record(f"poisson_nx{nx}_cg", cg, A, b, tol=1e-6))
    for nx in (1000, 2000):
        A = mu.create_convdiff_2d_csr(nx, nx)
        g = torch.Generator().manual_seed(0)
        xt = torch.randn(nx * nx, dtype=torch.float64, generator=g)
        b = A @ xt
        runs.append(record(f"convdiff_nx{nx}_bicgstab", bicgstab, A, b, tol=1e-6))
    with open(os.path.join(ROOT, "tests", "golden", "big_index.json"), "w") as f:
        json.dump({"generator": "oracle/gen_golden_big.py", "torch": torch.__version__,
                   "rhs": {"poisson": "ones", "convdiff": "A @ randn(n, generator=manual_seed(0)) computed with torch CSR matmul"},
                   "runs": runs}, f, indent=1)


if __name__ == "__main__":
    main()
