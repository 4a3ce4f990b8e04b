#!/usr/bin/env python3
"""Coded SpMV on the N = 4M Poisson matrix with the uniform tiles taken two rows per lane (hipk_spmv_sell_wide_kernel) or
one row per lane (HIPK_SPMV_SELL_NO_WIDE=1, read once per process): stand-alone SpMV, SpMV inside the CG loop, CG it/s,
GMRES(30) cycle.  Run once per setting."""
import os as _os; _os.environ.setdefault("HIPK_SPMV_NO_PLAN_CACHE", "1")  # this probe flips SpMV switches between launches
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd"))
import torch  # noqa: E402
from pytorch_sparse_solver import _hipk  # noqa: E402
from pytorch_sparse_solver.utils.matrix_utils import create_poisson_2d_csr  # noqa: E402

dev = torch.device("cuda", 0)
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
A = create_poisson_2d_csr(nx, nx, device=dev)
h = _hipk.CsrHandle(A.crow_indices(), A.col_indices(), A.values(), A.shape)
b = torch.ones(nx * nx, dtype=torch.float64, device=dev)
xr = torch.randn(nx * nx, dtype=torch.float64, device=dev, generator=torch.Generator(device=dev).manual_seed(0))
yr = torch.empty_like(xr)
for _ in range(20):
    _hipk.spmv(h, xr, out=yr)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(400):
    _hipk.spmv(h, xr, out=yr)
e1.record()
torch.cuda.synchronize()
alone = e0.elapsed_time(e1) / 400 * 1e3
x = torch.zeros_like(b)
_hipk.solve("cg", h, b, x, tol=1e-6, atol=0.0, maxiter=None)
torch.cuda.synchronize()
best = 0.0
for _ in range(3):
    x.zero_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    st = _hipk.solve("cg", h, b, x, tol=1e-6, atol=0.0, maxiter=None)
    torch.cuda.synchronize()
    best = max(best, st.iterations / (time.perf_counter() - t0))
x.zero_()
pst = _hipk.solve("cg", h, b, x, tol=1e-6, atol=0.0, maxiter=256, profile=True)
print(json.dumps({"no_wide": os.environ.get("HIPK_SPMV_SELL_NO_WIDE"), "nx": nx, "path": h.path(), "spmv_alone_us": round(alone, 2),
                  "spmv_in_cg_us": round(pst.spmv_ms_avg * 1e3, 2), "cg_it_s": round(best, 1), "iterations": st.iterations,
                  "x_sum": float(x.sum())}))
