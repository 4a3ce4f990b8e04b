"""Where does the fixed cost of a small solve go?  cProfile of module_a.cg on a 10x10 Poisson system
(handle cached) plus the stats side channel; run on the GPU box."""
import cProfile
import io
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..",
                                "pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd"))
from pytorch_sparse_solver import module_a  # noqa: E402
from pytorch_sparse_solver.utils import create_poisson_2d_csr  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    for nx in (10, 22, 100):
        A = create_poisson_2d_csr(nx, nx, device=dev)
        b = torch.ones(nx * nx, dtype=torch.float64, device=dev)
        for method in ("cg", "bicgstab", "gmres"):
            f = getattr(module_a, method)
            for _ in range(3):
                x, info = f(A, b, tol=1e-8)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            reps = 20
            for _ in range(reps):
                x, info = f(A, b, tol=1e-8)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            st = module_a.get_last_stats()
            print(f"n={nx*nx} {method}: {dt*1e3:.3f} ms/call, stats={st}", flush=True)
    A = create_poisson_2d_csr(10, 10, device=dev)
    b = torch.ones(100, dtype=torch.float64, device=dev)
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(50):
        module_a.cg(A, b, tol=1e-8)
    torch.cuda.synchronize()
    pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
    print(s.getvalue())


if __name__ == "__main__":
    main()
