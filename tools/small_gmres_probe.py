#!/usr/bin/env python3
"""BASELINE config 4 timing: gmres(restart=30) on the LDC pressure system, nx = 100 (n = 10^4), reference RHS of FVM step 0.
One launch per restart cycle (LDS-resident basis: default; one workgroup per chunk) against the multi-launch small-system path."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd")]
import numpy as np, torch
from pytorch_sparse_solver.module_a import gmres, get_last_stats
d = np.load(os.path.join(ROOT, "tests", "golden", "ldc_nx100_step0.npz"))
n = int(d["n"])
dev = "cuda:0"
for dt, tol in ((torch.float64, 1e-10), (torch.float32, 1e-5)):
    A = torch.sparse_csr_tensor(torch.from_numpy(d["crow"]).long(), torch.from_numpy(d["col"]).long(),
                                torch.from_numpy(d["val"]).to(dt), size=(n, n)).to(dev)
    b = torch.from_numpy(d["b"]).to(dev)
    for env in ({}, {"HIPK_GMRES_NO_LDS_CYCLE": "1"}, {"HIPK_GMRES_NO_CYCLE": "1"}):
        for k in ("HIPK_GMRES_NO_CYCLE", "HIPK_GMRES_NO_LDS_CYCLE"):
            os.environ.pop(k, None)
        os.environ.update(env)
        for rep in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            x, info = gmres(A, b, tol=tol, restart=30, maxiter=1000)
            torch.cuda.synchronize()
            dt_s = time.perf_counter() - t0
        st = get_last_stats()
        print(f"{str(dt):14s} {str(env):34s} solve {dt_s * 1e3:8.2f} ms  cycles {st.iterations}  ms/cycle {dt_s * 1e3 / st.iterations:.3f}  "
              f"info {info} relres {st.residual_norm / st.b_norm:.2e}", flush=True)
