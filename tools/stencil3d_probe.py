#!/usr/bin/env python3
"""7-point and 27-point 3-D stencils through every applicable SpMV path: constant coefficients (pair-coded), random
coefficients (offset-coded) and the general CSR kernels on the same matrices."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd"))
import numpy as np, torch
from pytorch_sparse_solver import _hipk
dev = "cuda:0"


def stencil3d(n, offs, random_vals):
    N = n ** 3
    idx = np.arange(N)
    i, j, k = idx // (n * n), (idx // n) % n, idx % n
    rows, cols, vals = [], [], []
    rng = np.random.default_rng(0)
    for q, (di, dj, dk) in enumerate(offs):
        ok = (i + di >= 0) & (i + di < n) & (j + dj >= 0) & (j + dj < n) & (k + dk >= 0) & (k + dk < n)
        rows.append(idx[ok]); cols.append(((i + di) * n * n + (j + dj) * n + (k + dk))[ok])
        vals.append(rng.standard_normal(ok.sum()) if random_vals else np.full(ok.sum(), 26.0 if (di, dj, dk) == (0, 0, 0) else -1.0))
    r, c, v = np.concatenate(rows), np.concatenate(cols), np.concatenate(vals)
    order = np.lexsort((c, r))
    r, c, v = r[order], c[order], v[order]
    crow = np.zeros(N + 1, dtype=np.int64); np.add.at(crow, r + 1, 1); crow = np.cumsum(crow)
    return crow, c.astype(np.int64), v, N


def timeit(h, x, y, reps=100):
    for _ in range(10): _hipk.spmv(h, x, out=y)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): _hipk.spmv(h, x, out=y)
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps * 1e3


o7 = [(0, 0, 0), (1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)]
o27 = [(a, b, c) for a in (-1, 0, 1) for b in (-1, 0, 1) for c in (-1, 0, 1)]
for name, n, offs in (("7-point 160^3", 160, o7), ("27-point 120^3", 120, o27)):
    for rv in (False, True):
        crow, col, val, N = stencil3d(n, offs, rv)
        h = _hipk.CsrHandle(torch.from_numpy(crow).to(dev), torch.from_numpy(col).to(dev), torch.from_numpy(val).to(dev), (N, N))
        x = torch.randn(N, dtype=torch.float64, device=dev); y = torch.empty_like(x)
        out = {"matrix": name, "values": "random" if rv else "constant", "n": N, "path": h.path(), "csr_MB": h.spmv_bytes() / 1e6,
               "format_MB": h.format_bytes() / 1e6}
        out["auto_us"] = timeit(h, x, y); ya = y.clone()
        h.set_path(plain_only=True)
        out["plain_path"] = h.path(); out["plain_us"] = timeit(h, x, y)
        out["bit_equal"] = bool(torch.equal(ya, y))
        out["auto_csr_TBps"] = h.spmv_bytes() / out["auto_us"] / 1e6; out["plain_TBps"] = h.spmv_bytes() / out["plain_us"] / 1e6
        print(json.dumps(out), flush=True)
