#!/usr/bin/env python3
"""Mid-size systems (16 k < n <= 260 k): time per CG / BiCGStab iteration and per GMRES(30) cycle, whole loop / solve in one
launch with the workgroups spread over the chip (default up to 32 reduction chunks = n <= 65536) against the launch sequences
(HIPK_NO_LDS_SPREAD=1)."""
import os, sys, time, torch
sys.path[:0] = ["pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd"]
from pytorch_sparse_solver.module_a import cg, gmres, bicgstab, get_last_stats
from pytorch_sparse_solver.utils.matrix_utils import create_poisson_2d_csr, create_convdiff_2d_csr
dev = "cuda:0"
for env in ({}, {"HIPK_NO_LDS_SPREAD": "1"}):
  os.environ.pop("HIPK_NO_LDS_SPREAD", None)
  os.environ.update(env)
  for nx in (128, 181, 256, 362):
      A = create_convdiff_2d_csr(nx, nx, device=dev); P = create_poisson_2d_csr(nx, nx, device=dev)
      n = A.shape[0]
      b = torch.ones(n, dtype=torch.float64, device=dev)
      for name, fn, M, kw in (("gmres", gmres, A, dict(tol=1e-12, restart=30, maxiter=10)), ("cg", cg, P, dict(tol=1e-12, maxiter=300)), ("bicgstab", bicgstab, A, dict(tol=1e-14, maxiter=150))):
          for _ in range(2): fn(M, b, **kw)
          torch.cuda.synchronize(); t0 = time.perf_counter()
          fn(M, b, **kw)
          torch.cuda.synchronize(); dt = time.perf_counter() - t0
          st = get_last_stats()
          print(f"{str(env):30s} n={n:7d} {name:8s} {dt*1e3:8.3f} ms  units {st.iterations}  per unit {dt*1e6/max(st.iterations,1):8.2f} us", flush=True)
