#!/usr/bin/env python3
"""Offset-coded SpMV layout vs the general CSR kernels on a variable-coefficient 5-point matrix (N = 4M)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd"))
import torch
from pytorch_sparse_solver import _hipk
from pytorch_sparse_solver.utils.matrix_utils import create_variable_diffusion_2d_csr
dev = torch.device("cuda", 0)
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
A = create_variable_diffusion_2d_csr(nx, nx, device=dev)
t0 = time.perf_counter(); h = _hipk.handle_for(A); torch.cuda.synchronize(); create_ms = (time.perf_counter() - t0) * 1e3
n = nx * nx
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(n, dtype=torch.float64, device=dev, generator=g); y = torch.empty_like(x)
b = torch.randn(n, dtype=torch.float64, device=dev, generator=g)
dinv = 1.0 / torch.zeros(n, dtype=torch.float64, device=dev).index_add_(0, torch.arange(n, device=dev), torch.ones(n, dtype=torch.float64, device=dev))
from pytorch_sparse_solver.module_a import JacobiPreconditioner
M = JacobiPreconditioner(A)
for plain in (True, False):
    h.set_path(plain_only=plain)
    for _ in range(20): _hipk.spmv(h, x, out=y)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): _hipk.spmv(h, x, out=y)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 200 * 1e3
    xs = torch.zeros_like(b); _hipk.solve("cg", h, b, xs, tol=1e-14, atol=0.0, maxiter=100)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    xs.zero_(); st = _hipk.solve("cg", h, b, xs, tol=1e-14, atol=0.0, maxiter=2000)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    xs.zero_(); t0 = time.perf_counter()
    st2 = _hipk.solve_pcg(h, M.dinv, b, xs, tol=1e-14, atol=0.0, maxiter=2000)
    torch.cuda.synchronize(); dt2 = time.perf_counter() - t0
    print(json.dumps({"path": h.path(), "create_ms": create_ms, "spmv_us": us, "csr_GBps": h.spmv_bytes() / us / 1e3,
                      "format_MB": h.format_bytes() / 1e6, "format_GBps": h.format_bytes() / us / 1e3,
                      "cg_it_per_s": st.iterations / dt, "pcg_it_per_s": st2.iterations / dt2}), flush=True)
