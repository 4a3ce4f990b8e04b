#!/usr/bin/env python3
"""GMRES(30) at N = 4M: A/B of the large-system kernel variants (env switches of hipk_gmres_solve_t) in one process.
Prints ms per restart cycle and checks that every variant returns the same bits."""
import os as _os; _os.environ.setdefault("HIPK_SPMV_NO_PLAN_CACHE", "1")  # this probe flips SpMV switches between launches
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd")]
import torch
from pytorch_sparse_solver.module_a import gmres, get_last_stats
from pytorch_sparse_solver.utils.matrix_utils import create_convdiff_2d_csr
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
A = create_convdiff_2d_csr(nx, nx, device="cuda:0")
b = torch.ones(nx * nx, dtype=torch.float64, device="cuda:0")
KEYS = ("HIPK_GMRES_NO_SWEEP", "HIPK_GM_MAP", "HIPK_GM_FOLD", "HIPK_GM_NRES", "HIPK_GM_SPEC", "HIPK_GMRES_NO_STREAM", "HIPK_GM_SWEEP3", "HIPK_SPMV_SELL_NO_WIDE")
variants = [a.split(",") for a in sys.argv[2:]] or [["HIPK_GMRES_NO_SWEEP=1"], []]
ref = None
for rep in range(2):
    for v in variants:
        for k in KEYS:
            os.environ.pop(k, None)
        for kv in v:
            if kv:
                k, val = kv.split("=")
                os.environ[k] = val
        gmres(A, b, tol=1e-6, restart=30, maxiter=1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        x, info = gmres(A, b, tol=1e-6, restart=30, maxiter=8)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        st = get_last_stats()
        if ref is None:
            ref = x.clone()
        print(f"{' '.join(v) or 'default':60s} ms/cycle {dt * 1e3 / st.iterations:7.3f} matvecs {st.matvecs} same_bits {bool(torch.equal(x, ref))}", flush=True)
