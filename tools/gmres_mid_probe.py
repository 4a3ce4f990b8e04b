#!/usr/bin/env python3
"""GMRES(30) per restart cycle on launch-bound mid-size systems (convection-diffusion nx^2, b = ones): the one-launch step loop
(hipk_gm_mid.h, "mid") against the launch sequence (HIPK_GMRES_MID=0), same process, alternating."""
import hashlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd"))
import torch
from pytorch_sparse_solver import _hipk
from pytorch_sparse_solver.utils.matrix_utils import create_convdiff_2d_csr
dev = torch.device("cuda", 0)
for nx in [int(a) for a in sys.argv[1:]] or (300, 400, 500, 600, 720):
    A = create_convdiff_2d_csr(nx, nx, device=dev)
    h = _hipk.handle_for(A)
    b = torch.ones(nx * nx, dtype=torch.float64, device=dev)
    for rep in range(2):
        for var in ("mid", "seq"):
            os.environ["HIPK_GMRES_MID"] = "1" if var == "mid" else "0"
            x = torch.zeros_like(b)
            _hipk.solve("gmres", h, b, x, tol=1e-12, atol=0.0, maxiter=2, restart=30)
            x.zero_()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            st = _hipk.solve("gmres", h, b, x, tol=1e-12, atol=0.0, maxiter=20, restart=30)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print(json.dumps({"n": nx * nx, "chunks": -(-nx * nx // 2048), "path": var, "cycles": st.iterations, "matvecs": st.matvecs,
                              "ms_per_cycle": round(dt / max(st.iterations, 1) * 1e3, 3),
                              "x_sha": hashlib.sha1(x.cpu().numpy().tobytes()).hexdigest()[:12]}), flush=True)
