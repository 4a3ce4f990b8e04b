#!/usr/bin/env python3
"""BiCGStab at N = 4M (convection-diffusion) for rocprofv3 kernel traces."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd")]
import torch
from pytorch_sparse_solver.module_a import bicgstab, get_last_stats
from pytorch_sparse_solver.utils.matrix_utils import create_convdiff_2d_csr
A = create_convdiff_2d_csr(2000, 2000, device="cuda:0")
g = torch.Generator(device="cuda:0").manual_seed(0)
b = torch.mv(A, torch.randn(4_000_000, dtype=torch.float64, device="cuda:0", generator=g))
bicgstab(A, b, tol=1e-6, maxiter=50)
torch.cuda.synchronize(); t0 = time.perf_counter()
x, info = bicgstab(A, b, tol=1e-6)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
st = get_last_stats()
print("iterations", st.iterations, "us/iter", dt / st.iterations * 1e6, "info", info)
