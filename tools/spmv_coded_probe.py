#!/usr/bin/env python3
"""A/B of the coded SpMV path (csrc/hipk_coded.h) against the plain CSR kernels on the N = 4M Poisson matrix:
stand-alone SpMV time, in-loop SpMV time and CG iterations/s, for 1/2/4 tiles per workgroup."""
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
nx = int(os.environ.get("NX", "2000"))
A = create_poisson_2d_csr(nx, nx, device=dev)
crow, col, val = A.crow_indices(), A.col_indices(), A.values()
b = torch.ones(nx * nx, dtype=torch.float64, device=dev)
xr = torch.randn(nx * nx, dtype=torch.float64, device=dev, generator=torch.Generator(device=dev).manual_seed(0))
yr = torch.empty_like(xr)


def spmv_us(h, reps=200):
    for _ in range(20):
        _hipk.spmv(h, xr, out=yr)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        _hipk.spmv(h, xr, out=yr)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def cg_rate(h):
    x = torch.zeros_like(b)
    _hipk.solve("cg", h, b, x, tol=1e-6, atol=0.0, maxiter=None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    x.zero_()
    st = _hipk.solve("cg", h, b, x, tol=1e-6, atol=0.0, maxiter=None)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    x.zero_()
    pst = _hipk.solve("cg", h, b, x, tol=1e-6, atol=0.0, maxiter=256, profile=True)
    return st.iterations / dt, st.iterations, pst.spmv_ms_avg * 1e3


rows = []
for label, plain, layout, loop in (("plain", True, "sell", "1"), ("coded csr", False, "csr", "1"),
                                   ("coded sell loop x1", False, "sell", "1"), ("coded sell loop x2", False, "sell", "2")):
    os.environ["HIPK_SPMV_CODED_LAYOUT"] = layout
    os.environ["HIPK_SPMV_SELL_LOOP"] = loop
    t0 = time.perf_counter()
    h = _hipk.CsrHandle(crow, col, val, A.shape)
    torch.cuda.synchronize()
    create_ms = (time.perf_counter() - t0) * 1e3
    h.set_path(plain_only=plain)
    us = spmv_us(h)
    rate, its, inloop = cg_rate(h)
    rows.append({"variant": label, "path": h.path(), "create_ms": create_ms, "spmv_us": us,
                 "algorithmic_GBps": h.spmv_bytes() / us / 1e3, "format_GBps": h.format_bytes() / us / 1e3,
                 "format_MB": h.format_bytes() / 1e6, "cg_it_per_s": rate, "cg_iterations": its,
                 "spmv_in_loop_us": inloop})
    print(json.dumps(rows[-1]), flush=True)
    h.close()
