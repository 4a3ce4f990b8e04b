#!/usr/bin/env python3
"""Coded SpMV on Poisson grids whose reduction chunks span several grid lines (nx = 4000: chunk = 32 tiles = 2 grid lines,
nx = 8000 = BASELINE config 5 on one GPU: chunk = 128 tiles = 4 grid lines): the two-rows-per-lane kernel with every workgroup
walking its own chunk (HIPK_SPMV_SELL_STRIDED=0) against one workgroup per group of 4 tiles on an ordinary grid (=1) and the
library's own choice (unset), in ONE process (the switch is read per launch); stand-alone SpMV, SpMV inside the CG loop, CG
time per iteration, x of 200 iterations bitwise equal.  HIPK_SPMV_SELL_CHUNKED=0 in the environment measures the
one-row-per-lane persistent kernel instead."""
import os as _os; _os.environ.setdefault("HIPK_SPMV_NO_PLAN_CACHE", "1")  # this probe flips SpMV switches between launches
import hashlib
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
for spec in (sys.argv[1:] or ["8000"]):   # "8000" | "f32:8000" (fp32 storage) | "var:4000" (variable coefficients: offset-coded form)
    kind, nx = (spec.split(":") + [None])[:2] if ":" in spec else ("f64", spec)
    nx = int(nx)
    if kind == "var":
        from pytorch_sparse_solver.utils.matrix_utils import create_variable_diffusion_2d_csr
        A = create_variable_diffusion_2d_csr(nx, nx, device=dev)
    else:
        A = create_poisson_2d_csr(nx, nx, device=dev)
    dt_ = torch.float32 if kind == "f32" else torch.float64
    vals = A.values().to(dt_)
    h = _hipk.CsrHandle(A.crow_indices(), A.col_indices(), vals, A.shape)
    n = nx * nx
    b = torch.ones(n, dtype=dt_, device=dev)
    xr = torch.randn(n, dtype=dt_, device=dev, generator=torch.Generator(device=dev).manual_seed(0))
    yr = torch.empty_like(xr)
    reps = max(20, int(4e8 // n))
    digests = {}
    for strided in ("0", "1", None):
        if strided is None:
            os.environ.pop("HIPK_SPMV_SELL_STRIDED", None)
        else:
            os.environ["HIPK_SPMV_SELL_STRIDED"] = strided
        for _ in range(5):
            _hipk.spmv(h, xr, out=yr)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            _hipk.spmv(h, xr, out=yr)
        e1.record()
        torch.cuda.synchronize()
        alone = e0.elapsed_time(e1) / reps * 1e3
        kern_alone = _hipk.CsrHandle.last_spmv_kernel()
        ysum = hashlib.sha1(yr.cpu().numpy().tobytes()).hexdigest()[:12]
        x = torch.zeros_like(b)
        _hipk.solve("cg", h, b, x, tol=1e-12, atol=0.0, maxiter=10)
        x.zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        st = _hipk.solve("cg", h, b, x, tol=1e-12, atol=0.0, maxiter=200)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        xd = hashlib.sha1(x.cpu().numpy().tobytes()).hexdigest()[:12]
        x.zero_()
        pst = _hipk.solve("cg", h, b, x, tol=1e-12, atol=0.0, maxiter=128, profile=True)
        kern = _hipk.CsrHandle.last_spmv_kernel()
        digests[strided] = (ysum, xd)
        print(json.dumps({"nx": nx, "kind": kind, "path": h.path(), "strided": strided, "chunked_env": os.environ.get("HIPK_SPMV_SELL_CHUNKED"),
                          "kernel": kern, "kernel_alone": kern_alone, "format_MB": round(h.format_bytes() / 1e6, 1) if hasattr(h, "format_bytes") else None,
                          "spmv_alone_us": round(alone, 2), "spmv_in_cg_us": round(pst.spmv_ms_avg * 1e3, 2),
                          "cg_us_per_iter": round(dt / st.iterations * 1e6, 2), "y_sha": ysum, "x_sha": xd}), flush=True)
    print("nx", nx, "bitwise equal across settings:", len(set(digests.values())) == 1, flush=True)
    del A, h, b, xr, yr, x, vals
    _hipk.clear_cache()
    torch.cuda.empty_cache()
