#!/usr/bin/env python3
"""BASELINE config 5's system at its full size (5-point Poisson 8000 x 8000, N = 64,000,000, b = ones, cg(tol=1e-6)) run by THE
REFERENCE (imported from /root/reference/src) on CPU -> tests/golden/config5_index.json: first with the loop cut at 150
iterations (3.5 minutes on 6 threads), then -- with --full -- the whole solve (13,429 iterations at 1.4 s each: five hours and
46 GB; not run for the committed fixture).  Only scalars and 16 sampled entries of x are stored
(SURVEY 8c).  Build container only.
Usage: PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_config5.py [--full]
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

from pytorch_sparse_solver.module_a import cg  # noqa: E402  (the REFERENCE package: first on sys.path)
import pytorch_sparse_solver  # noqa: E402

assert pytorch_sparse_solver.__file__.startswith("/root/reference"), pytorch_sparse_solver.__file__

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


def record(name, A, b, **kw):
    op = Counting(A)
    t0 = time.time()
    x, info = cg(op, b, **kw)
    dt = time.time() - t0
    n = b.numel()
    nx = int(round(n ** 0.5))
    # 8 pseudo-random rows + 8 near the boundary, where x varies after a few iterations (the interior is still flat then)
    idx = [(i * 2654435761) % n for i in range(8)] + [i * nx + j for i, j in ((0, 0), (0, 17), (3, nx // 2), (40, nx - 1), (149, 1),
                                                                             (nx - 1, nx - 1), (nx - 10, 13), (nx // 2, 3))]
    res = torch.norm(b - A @ x).item()
    w = ((torch.arange(n, dtype=torch.int64) * 2654435761) % 1000).to(torch.float64) / 1000.0   # a position-sensitive functional
    out = {"case": name, "n": n, "kwargs": kw, "info": int(info), "matvecs": op.count, "relres": res / torch.norm(b).item(),
           "x_norm": torch.norm(x).item(), "x_dot_w": torch.dot(x, w).item(), "w": "((arange(n) * 2654435761) % 1000) / 1000",
           "sample_idx": idx, "sample_x": [x[i].item() for i in idx], "seconds": dt}
    print(json.dumps(out), flush=True)
    return out


def dump(runs):
    with open(os.path.join(ROOT, "tests", "golden", "config5_index.json"), "w") as f:
        json.dump({"generator": "oracle/gen_golden_config5.py", "torch": torch.__version__, "rhs": "ones",
                   "matrix": "create_poisson_2d_csr(8000, 8000) (bit-identical to the reference's loop builder, tests/test_matrix_utils)",
                   "runs": runs}, f, indent=1)


def main():
    torch.set_num_threads(int(os.environ.get("GEN_THREADS", "6")))
    nx = 8000
    A = mu.create_poisson_2d_csr(nx, nx)
    b = torch.ones(nx * nx, dtype=torch.float64)
    runs = [record("poisson_nx8000_cg_maxiter150", A, b, tol=1e-6, maxiter=150)]
    dump(runs)
    if "--full" not in sys.argv:
        return
    runs.append(record("poisson_nx8000_cg", A, b, tol=1e-6))
    dump(runs)


if __name__ == "__main__":
    main()
