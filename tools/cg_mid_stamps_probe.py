#!/usr/bin/env python3
"""Where does an iteration of hipk_cg_mid_kernel go?  Run with HIPK_LIB_PATH=.../libhipk_stamps.so (make -C csrc stamps): thread 0 of
every workgroup sums the constant 100 MHz clock between its phase boundaries over the iterations of one launch.  Prints, per
phase, the average per iteration in microseconds: median / min / max over the workgroups."""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd"))
import numpy as np
import torch
from pytorch_sparse_solver import _hipk
from pytorch_sparse_solver.utils.matrix_utils import create_poisson_2d_csr
L = _hipk.lib()
if not hasattr(L, "hipk_debug_mid_stamps"):
    raise SystemExit("needs the stamps twin: HIPK_LIB_PATH=.../_lib/libhipk_stamps.so (make -C csrc stamps)")
NS = 12
NAMES = ["A p + tile sums + barrier", "chunk partial of <p,Ap> published", "poll <p,Ap> partials", "block fold", "r, x, r published + barrier",
         "<r,r> chains", "block fold + publish", "poll halo r", "poll <r,r> partials", "block fold", "p window + barrier"]
dev = torch.device("cuda", 0)
for nx in [int(a) for a in sys.argv[1:]] or [300, 500, 720, 1000]:
    A = create_poisson_2d_csr(nx, nx, device=dev)
    h = _hipk.handle_for(A)
    n = nx * nx
    b = torch.ones(n, dtype=torch.float64, device=dev)
    for rep in range(2):
        x = torch.zeros_like(b)
        st = _hipk.solve("cg", h, b, x, tol=1e-12, atol=0.0, maxiter=1000)
    buf = (ctypes.c_ulonglong * (512 * NS))()
    L.hipk_debug_mid_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    _hipk._check(L.hipk_debug_mid_stamps(buf, len(buf)), "stamps")
    g = -(-n // 2048)
    t = np.frombuffer(buf, dtype=np.uint64).reshape(512, NS)[:g].astype(np.float64) / 100.0 / st.iterations
    print(json.dumps({"n": n, "chunks": g, "iterations": st.iterations, "us_per_iteration_sum": round(float(np.median(t.sum(axis=1))), 2),
                      "phases": {NAMES[k]: [round(float(np.median(t[:, k])), 2), round(float(t[:, k].min()), 2), round(float(t[:, k].max()), 2)]
                                 for k in range(len(NAMES))}}), flush=True)
