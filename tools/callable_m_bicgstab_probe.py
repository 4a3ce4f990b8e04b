#!/usr/bin/env python3
"""bicgstab() and gmres(30) with a callable preconditioner at N = 4M (convection-diffusion): device-resident Jacobi vs
the callback variant of the same loop; fixed 200 iterations / 6 restart cycles."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd")]
import torch
from pytorch_sparse_solver.module_a import JacobiPreconditioner, bicgstab, get_last_stats, gmres
from pytorch_sparse_solver.utils.matrix_utils import create_convdiff_2d_csr
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
A = create_convdiff_2d_csr(nx, nx, device="cuda:0")
b = torch.ones(nx * nx, dtype=torch.float64, device="cuda:0")
J = JacobiPreconditioner(A)
dinv = J.dinv
for name, M in (("jacobi_device_resident", J), ("callable_in_the_device_loop", lambda v: dinv * v)):
    bicgstab(A, b, M=M, tol=1e-12, maxiter=10)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    x, info = bicgstab(A, b, M=M, tol=1e-12, maxiter=200)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = get_last_stats()
    print(f"{name}: {st.iterations} iterations, {dt * 1e6 / st.iterations:.1f} us/iteration, method {st.method}", flush=True)
for name, M in (("gmres30 jacobi_device_resident", J), ("gmres30 callable_in_the_device_loop", lambda v: dinv * v)):
    gmres(A, b, M=M, tol=1e-12, restart=30, maxiter=1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    x, info = gmres(A, b, M=M, tol=1e-12, restart=30, maxiter=6)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = get_last_stats()
    print(f"{name}: {st.iterations} cycles, {dt * 1e3 / st.iterations:.2f} ms/cycle, method {st.method}", flush=True)
