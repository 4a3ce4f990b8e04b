#!/usr/bin/env python3
"""CG per iteration on 3-D 7-point Poisson grids m^3: the one-launch loop (windows as lists of 256-column tiles: own rows, the
planes below and above) against the launch sequences (HIPK_CG_MID=0), same process, alternating."""
import hashlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd"))
import numpy as np, scipy.sparse as sp, torch
from pytorch_sparse_solver import _hipk
dev = torch.device("cuda", 0)
for m in [int(a) for a in sys.argv[1:]] or (40, 48, 56, 64, 72, 80):
    T = sp.diags([-1.0, 2.0, -1.0], [-1, 0, 1], shape=(m, m)); I = sp.identity(m)
    M = (sp.kron(sp.kron(T, I), I) + sp.kron(sp.kron(I, T), I) + sp.kron(sp.kron(I, I), T)).tocsr(); M.sort_indices()
    A = torch.sparse_csr_tensor(torch.from_numpy(M.indptr.astype(np.int64)), torch.from_numpy(M.indices.astype(np.int64)),
                                torch.from_numpy(M.data), size=M.shape).to(dev)
    h = _hipk.handle_for(A)
    n = m ** 3
    b = torch.ones(n, dtype=torch.float64, device=dev)
    for rep in range(2):
        for var in ("mid", "seq"):
            os.environ["HIPK_CG_MID"] = "1" if var == "mid" else "0"
            x = torch.zeros_like(b)
            _hipk.solve("cg", h, b, x, tol=1e-12, atol=0.0, maxiter=30)
            x.zero_()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            st = _hipk.solve("cg", h, b, x, tol=1e-12, atol=0.0, maxiter=500)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print(json.dumps({"grid": f"{m}^3", "n": n, "chunks": -(-n // 2048), "path": var, "iterations": st.iterations,
                              "us_per_iteration": round(dt / st.iterations * 1e6, 2),
                              "x_sha": hashlib.sha1(x.cpu().numpy().tobytes()).hexdigest()[:12]}), flush=True)
