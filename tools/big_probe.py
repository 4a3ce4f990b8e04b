#!/usr/bin/env python3
"""BASELINE config 5's matrix on ONE GPU (N = 64M, nnz = 320M): does everything hold at that size?  50 CG iterations on
the coded path and on the general CSR kernels, bitwise equal; it/s for both."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd"))
import torch
from pytorch_sparse_solver import _hipk
from pytorch_sparse_solver.utils.matrix_utils import create_poisson_2d_csr
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
A = create_poisson_2d_csr(nx, nx, device="cuda:0")
n = nx * nx
b = torch.ones(n, dtype=torch.float64, device="cuda:0")
t0 = time.perf_counter(); h = _hipk.handle_for(A); torch.cuda.synchronize(); create = time.perf_counter() - t0
res = {}
print("HIPK_SPMV_SELL_DEPTH =", os.environ.get("HIPK_SPMV_SELL_DEPTH", "auto"))
for plain in (False, True):
    h.set_path(plain_only=plain)
    x = torch.zeros_like(b)
    _hipk.solve("cg", h, b, x, tol=1e-12, atol=0.0, maxiter=10)
    x.zero_(); torch.cuda.synchronize(); t0 = time.perf_counter()
    st = _hipk.solve("cg", h, b, x, tol=1e-12, atol=0.0, maxiter=200)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    res[plain] = x.clone()
    print(json.dumps({"n": n, "nnz": h.nnz, "path": h.path(), "create_s": create, "iterations": st.iterations,
                      "us_per_iter": dt / st.iterations * 1e6, "it_per_s_in_4M_units": st.iterations / dt * n / 4e6,
                      "chunk": int(_hipk.lib().hipk_chunk_size(n))}), flush=True)
print("bitwise equal:", bool(torch.equal(res[False], res[True])))
