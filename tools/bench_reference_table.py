#!/usr/bin/env python3
"""The reference's only published table (README.md:628-634) re-measured on this build:
`python src/run.py --benchmark --quick` = dense 5-point Poisson of size floor(sqrt(n))^2 for n in (100, 200, 500),
RHS = A randn, solver.solve(A, b, method, backend='module_a', tol=1e-8, maxiter=1000), one warm-up, mean of 2 runs
with a device synchronise around each (benchmark.py:110-138, 221-248, 269).  Prints one JSON line per cell."""
import json, math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd")]
import torch
from pytorch_sparse_solver import SparseSolver
from pytorch_sparse_solver.utils.matrix_utils import create_poisson_2d_sparse_coo

PUBLISHED_MS = {("cg", 100): 23.4, ("cg", 200): 68.3, ("cg", 500): 46.2,
                ("gmres", 100): 344.5, ("gmres", 200): 355.0, ("gmres", 500): 515.7}   # RTX 4090, README.md:632-634


def main():
    dev = "cuda:0"
    solver = SparseSolver()
    for n_req in (100, 200, 500):
        g = int(math.isqrt(n_req))
        A = create_poisson_2d_sparse_coo(g, g, device=dev).to_dense()     # the reference benchmark uses DENSE matrices
        gen = torch.Generator().manual_seed(n_req)
        b = A @ torch.randn(g * g, dtype=torch.float64, generator=gen).to(dev)
        for method in ("cg", "bicgstab", "gmres"):
            solver.solve(A, b, method=method, backend="module_a", tol=1e-8, maxiter=1000)      # warm-up
            times = []
            for _ in range(2):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                x, res = solver.solve(A, b, method=method, backend="module_a", tol=1e-8, maxiter=1000)
                torch.cuda.synchronize(); times.append(time.perf_counter() - t0)
            ms = 1e3 * sum(times) / len(times)
            pub = PUBLISHED_MS.get((method, n_req))
            print(json.dumps({"n_requested": n_req, "n": g * g, "method": method, "ms": ms, "converged": res.converged,
                              "residual": res.residual, "published_rtx4090_ms": pub,
                              "speedup_vs_published": (pub / ms) if pub else None}), flush=True)


if __name__ == "__main__":
    main()
