#!/usr/bin/env python3
"""Where does a restart cycle of hipk_gm_mid_kernel go?  Run with HIPK_LIB_PATH=.../libhipk_stamps.so (make -C csrc stamps): thread 0
of every workgroup sums the constant 100 MHz clock between its phase boundaries over the steps of one launch (= one GMRES(30)
cycle).  Prints, per phase, microseconds per cycle: median / min / max over the workgroups."""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd"))
import numpy as np
import torch
from pytorch_sparse_solver import _hipk
from pytorch_sparse_solver.utils.matrix_utils import create_convdiff_2d_csr
L = _hipk.lib()
if not hasattr(L, "hipk_debug_gm_mid_stamps"):
    raise SystemExit("needs the stamps twin: HIPK_LIB_PATH=.../_lib/libhipk_stamps.so (make -C csrc stamps)")
NS = 14
NAMES = ["A v + tile sums + barrier", "<V_j,w> chains + barrier", "chunk trees, partials out + barrier", "poll the partials + barrier",
         "column trees -> h + barrier", "q = w - V h + barrier", "<q,q> chains + barrier", "tree, out, poll <q,q> (+ ||Av||^2)", "fold + barrier",
         "H column, Givens + barrier", "poll halo v + barrier", "second-pass decision", "v = q/||q||: scaling, stores"]
dev = torch.device("cuda", 0)
for nx in [int(a) for a in sys.argv[1:]] or [300, 500, 720]:
    A = create_convdiff_2d_csr(nx, nx, device=dev)
    h = _hipk.handle_for(A)
    n = nx * nx
    b = torch.ones(n, dtype=torch.float64, device=dev)
    for rep in range(2):
        x = torch.zeros_like(b)
        st = _hipk.solve("gmres", h, b, x, tol=1e-12, atol=0.0, maxiter=3, restart=30)
    buf = (ctypes.c_ulonglong * (256 * NS))()
    L.hipk_debug_gm_mid_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    _hipk._check(L.hipk_debug_gm_mid_stamps(buf, len(buf)), "stamps")
    g = -(-n // 2048)
    t = np.frombuffer(buf, dtype=np.uint64).reshape(256, NS)[:g].astype(np.float64) / 100.0
    print(json.dumps({"n": n, "chunks": g, "us_per_cycle_sum": round(float(np.median(t.sum(axis=1))), 1),
                      "phases": {NAMES[k]: [round(float(np.median(t[:, k])), 1), round(float(t[:, k].min()), 1), round(float(t[:, k].max()), 1)]
                                 for k in range(len(NAMES))}}), flush=True)
