#!/usr/bin/env python3
"""Launch-bound CG: 5-point Poisson n = 10^4 (100 x 100) and the LDC pressure matrix nx = 100, b = ones; the whole loop in one
launch (default) against three launches per iteration (HIPK_CG_NO_LDS_LOOP=1)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd")]
import torch
from pytorch_sparse_solver.module_a import bicgstab, cg, get_last_stats
from pytorch_sparse_solver.utils.matrix_utils import create_convdiff_2d_csr, create_poisson_2d_csr
dev = "cuda:0"
for name, A in (("poisson 100x100", create_poisson_2d_csr(100, 100, device=dev)), ("poisson 128x128", create_poisson_2d_csr(128, 128, device=dev))):
    n = A.shape[0]
    b = torch.ones(n, dtype=torch.float64, device=dev)
    for env in ({}, {"HIPK_CG_NO_LDS_LOOP": "1"}):
        os.environ.pop("HIPK_CG_NO_LDS_LOOP", None)
        os.environ.update(env)
        for rep in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            x, info = cg(A, b, tol=1e-10)
            torch.cuda.synchronize()
            dt_s = time.perf_counter() - t0
        st = get_last_stats()
        print(f"{name:16s} {str(env):32s} solve {dt_s * 1e3:7.3f} ms  iterations {st.iterations}  us/iteration {dt_s * 1e6 / max(st.iterations, 1):6.2f}  "
              f"info {info} relres {st.residual_norm / st.b_norm:.2e}", flush=True)

A = create_convdiff_2d_csr(100, 100, device=dev)
n = A.shape[0]
b = A @ torch.randn(n, dtype=torch.float64, device=dev, generator=torch.Generator(device=dev).manual_seed(0))
for env in ({}, {"HIPK_BICGSTAB_NO_LDS_LOOP": "1"}):
    os.environ.pop("HIPK_BICGSTAB_NO_LDS_LOOP", None)
    os.environ.update(env)
    for rep in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        x, info = bicgstab(A, b, tol=1e-10)
        torch.cuda.synchronize()
        dt_s = time.perf_counter() - t0
    st = get_last_stats()
    print(f"{'convdiff 100x100':16s} {str(env):36s} bicgstab solve {dt_s * 1e3:7.3f} ms  iterations {st.iterations}  "
          f"us/iteration {dt_s * 1e6 / max(st.iterations, 1):6.2f}  info {info} relres {st.residual_norm / st.b_norm:.2e}", flush=True)
