#!/usr/bin/env python3
"""Where a tiny solve's time goes: wall time per cg()/gmres()/bicgstab() call through the public API against the device time
the C solve measures with events (hipk_stats.solve_ms)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd")]
import torch
from pytorch_sparse_solver.module_a import bicgstab, cg, gmres, get_last_stats
from pytorch_sparse_solver.utils.matrix_utils import create_poisson_2d_csr
dev = "cuda:0"
for nx in (10, 22, 100):
    A = create_poisson_2d_csr(nx, nx, device=dev)
    n = A.shape[0]
    b = torch.ones(n, dtype=torch.float64, device=dev)
    for name, fn, kw in (("cg", cg, dict(tol=1e-8)), ("bicgstab", bicgstab, dict(tol=1e-8)), ("gmres", gmres, dict(tol=1e-8, restart=30))):
        for _ in range(5):
            fn(A, b, **kw)
        torch.cuda.synchronize()
        reps, dev_ms = 100, 0.0
        t0 = time.perf_counter()
        for _ in range(reps):
            x, info = fn(A, b, **kw)
            dev_ms += get_last_stats().solve_ms
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / reps * 1e3
        st = get_last_stats()
        print(f"n={n:6d} {name:9s} wall {wall:7.3f} ms per call | device (events) {dev_ms / reps:7.3f} ms | iterations {st.iterations} info {info}", flush=True)
