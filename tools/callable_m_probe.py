#!/usr/bin/env python3
"""cg() with a callable preconditioner at N = 4M: the fused step-API path against the generic torch-op path
(HIPK_CG_CALLABLE_M=0) and against the device-resident Jacobi solve; fixed 300 iterations."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd")]
import torch
from pytorch_sparse_solver.module_a import JacobiPreconditioner, cg, get_last_stats
from pytorch_sparse_solver.utils.matrix_utils import create_variable_diffusion_2d_csr
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
A = create_variable_diffusion_2d_csr(nx, nx, device="cuda:0")
b = torch.randn(nx * nx, dtype=torch.float64, device="cuda:0", generator=torch.Generator(device="cuda:0").manual_seed(1))
J = JacobiPreconditioner(A)
dinv = J.dinv
for name, M, env in (("jacobi_device_resident", J, "1"), ("callable_fused", lambda v: dinv * v, "1"),
                     ("callable_generic_torch_ops", lambda v: dinv * v, "0")):
    os.environ["HIPK_CG_CALLABLE_M"] = env
    cg(A, b, M=M, tol=1e-12, maxiter=20)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    x, info = cg(A, b, M=M, tol=1e-12, maxiter=300)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = get_last_stats()
    print(f"{name}: {st.iterations} iterations, {dt * 1e6 / st.iterations:.1f} us/iteration, method {st.method}", flush=True)

# matrix-free operator (callable A) on device vectors: fused vector kernels + device stop word vs the generic torch-op path
from pytorch_sparse_solver import _hipk
h = _hipk.handle_for(A)
for name, env in (("matrix_free_fused", "1"), ("matrix_free_generic_torch_ops", "0")):
    os.environ["HIPK_CG_MATRIX_FREE"] = env
    cg(lambda v: _hipk.spmv(h, v), b, tol=1e-12, maxiter=20)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    x, info = cg(lambda v: _hipk.spmv(h, v), b, tol=1e-12, maxiter=300)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = get_last_stats()
    print(f"{name}: {st.iterations} iterations, {dt * 1e6 / st.iterations:.1f} us/iteration, method {st.method}", flush=True)
