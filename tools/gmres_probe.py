#!/usr/bin/env python3
"""GMRES(30) at N = 4M (convection-diffusion): a few restart cycles, for rocprofv3 kernel traces."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd")]
import torch
from pytorch_sparse_solver.module_a import gmres, get_last_stats
from pytorch_sparse_solver.utils.matrix_utils import create_convdiff_2d_csr, create_ldc_pressure_csr
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
method = sys.argv[2] if len(sys.argv) > 2 else "batched"
A = create_convdiff_2d_csr(nx, nx, device="cuda:0")
b = torch.ones(nx * nx, dtype=torch.float64, device="cuda:0")
gmres(A, b, tol=1e-6, restart=30, maxiter=2, solve_method=method)
torch.cuda.synchronize()
t0 = time.perf_counter()
x, info = gmres(A, b, tol=1e-6, restart=30, maxiter=10, solve_method=method)
torch.cuda.synchronize()
st = get_last_stats()
print("cycles", st.iterations, "ms/cycle", (time.perf_counter() - t0) * 1e3 / st.iterations, "matvecs", st.matvecs)
