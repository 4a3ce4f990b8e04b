#!/usr/bin/env python3
"""BiCGStab at N = 4M (convection-diffusion, BASELINE config 3): a fixed number of iterations, for rocprofv3
kernel traces.  usage: bicgstab_probe.py [nx] [maxiter]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd")]
import torch
from pytorch_sparse_solver.module_a import bicgstab, get_last_stats
from pytorch_sparse_solver.utils.matrix_utils import create_convdiff_2d_csr
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
maxiter = int(sys.argv[2]) if len(sys.argv) > 2 else 500
A = create_convdiff_2d_csr(nx, nx, device="cuda:0")
b = torch.ones(nx * nx, dtype=torch.float64, device="cuda:0")
bicgstab(A, b, tol=1e-12, maxiter=20)
torch.cuda.synchronize()
t0 = time.perf_counter()
x, info = bicgstab(A, b, tol=1e-12, maxiter=maxiter)
torch.cuda.synchronize()
st = get_last_stats()
dt = time.perf_counter() - t0
print("iterations", st.iterations, "us/iteration", dt * 1e6 / st.iterations, "matvecs", st.matvecs)
