#!/usr/bin/env python3
"""CG per iteration on launch-bound mid-size systems (5-point Poisson, 65 k < n <= 524 k and the sizes around): two launches per
iteration (hipk_cg2_*) against three (HIPK_CG_TWO_LAUNCH=0) and against the one-launch loop (hipk_cg_mid.h, "mid"), same
process, alternating."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd"))
import torch
from pytorch_sparse_solver import _hipk
from pytorch_sparse_solver.utils.matrix_utils import create_poisson_2d_csr
dev = torch.device("cuda", 0)
VARIANTS = sys.argv[1].split(",") if len(sys.argv) > 1 else ("mid", "mid-noxcd", "1", "0")
DT = torch.float32 if (len(sys.argv) > 2 and sys.argv[2] == "f32") else torch.float64   # fp32 storage
for nx in (200, 300, 400, 500, 600, 720, 1000):
    A = create_poisson_2d_csr(nx, nx, device=dev)
    if DT == torch.float32:
        A = torch.sparse_csr_tensor(A.crow_indices(), A.col_indices(), A.values().float(), size=A.shape)
    h = _hipk.handle_for(A)
    b = torch.ones(nx * nx, dtype=DT, device=dev)
    for rep in range(2):
        for two in VARIANTS:
            os.environ["HIPK_CG_MID"] = "1" if two.startswith("mid") else "0"
            os.environ["HIPK_CG_MID_XCD"] = "0" if two == "mid-noxcd" else "1"
            os.environ["HIPK_CG_MID_STRIDE"] = two[5:] if two.startswith("mid-s") else "16"
            os.environ["HIPK_CG_TWO_LAUNCH"] = "1" if two.startswith("mid") else two
            x = torch.zeros_like(b)
            _hipk.solve("cg", h, b, x, tol=1e-12, atol=0.0, maxiter=50)
            x.zero_()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            st = _hipk.solve("cg", h, b, x, tol=1e-12, atol=0.0, maxiter=1500)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print(json.dumps({"n": nx * nx, "chunks": -(-nx * nx // 2048), "two_launch": two, "iterations": st.iterations,
                              "us_per_iteration": round(dt / st.iterations * 1e6, 2), "x_sha": __import__("hashlib").sha1(x.cpu().numpy().tobytes()).hexdigest()[:12]}), flush=True)
