#!/usr/bin/env python3
"""CG direction step with the streaming policy (N = 16 M ... 64 M): one workgroup per chunk (HIPK_CG_FLAT_DIRECTION=0) against the
scalars launch + flat grid, alternating in ONE process on the same vectors (the switch is read per solve; the caching allocator
hands back the same blocks, so the physical placement -- which alone moves this kernel by 15 % -- is the same for both)."""
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
for nx in [int(v) for v in (sys.argv[1:] or ["8000"])]:
    A = create_poisson_2d_csr(nx, nx, device=dev)
    h = _hipk.CsrHandle(A.crow_indices(), A.col_indices(), A.values(), A.shape)
    n = nx * nx
    b = torch.ones(n, dtype=torch.float64, device=dev)
    x = torch.zeros_like(b)
    _hipk.solve("cg", h, b, x, tol=1e-12, atol=0.0, maxiter=10)
    ref = None
    for rep in range(3):
        for flat in ("0", "1"):
            os.environ["HIPK_CG_FLAT_DIRECTION"] = flat
            x.zero_()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            st = _hipk.solve("cg", h, b, x, tol=1e-12, atol=0.0, maxiter=200)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            xs = x.clone()
            x.zero_()
            pst = _hipk.solve("cg", h, b, x, tol=1e-12, atol=0.0, maxiter=128, profile=3)
            if ref is None:
                ref = xs
            print(json.dumps({"nx": nx, "flat": flat, "rep": rep, "cg_us_per_iter": round(dt / st.iterations * 1e6, 1),
                              "direction_step_us": round(pst.spmv_ms_avg * 1e3, 1), "bitwise_equal_to_first": bool(torch.equal(xs, ref))}), flush=True)
    del A, h, b, x, ref, xs
    _hipk.clear_cache()
    torch.cuda.empty_cache()
