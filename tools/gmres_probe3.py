#!/usr/bin/env python3
"""GMRES at N = 4 M (convection-diffusion, b = ones): ms per restart cycle for restart 30 / 50 / 100, an A/B of one environment
switch (argv[2], default HIPK_GM_MD_WIDE: multi-dot with one column group per workgroup against groups of eight;
HIPK_GM_SPLIT_NORM: normalise step as scalars launch + flat scale kernel against the one-kernel form), same process, alternating;
x must be bit-identical between the two."""
import hashlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd")]
import torch
from pytorch_sparse_solver.module_a import gmres, get_last_stats
from pytorch_sparse_solver.utils.matrix_utils import create_convdiff_2d_csr
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
switch = sys.argv[2] if len(sys.argv) > 2 else "HIPK_GM_MD_WIDE"
A = create_convdiff_2d_csr(nx, nx, device="cuda:0")
b = torch.ones(nx * nx, dtype=torch.float64, device="cuda:0")
shapes = ((30, 8), (50, 5), (100, 3)) if nx >= 2000 else ((30, 8),)
for restart, cycles in shapes:
    for rep in range(2):
        for wide in ("1", "0"):
            os.environ[switch] = wide
            gmres(A, b, tol=1e-12, restart=restart, maxiter=1)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            x, info = gmres(A, b, tol=1e-12, restart=restart, maxiter=cycles)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            st = get_last_stats()
            print(json.dumps({"nx": nx, "restart": restart, "switch": switch, "value": wide, "cycles": st.iterations, "ms_per_cycle": dt * 1e3 / st.iterations,
                              "matvecs": st.matvecs, "x_sha": hashlib.sha1(x.cpu().numpy().tobytes()).hexdigest()[:12]}), flush=True)
