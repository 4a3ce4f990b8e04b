#!/usr/bin/env python3
"""What the non-uniform tiles cost the two-rows-per-lane SpMV: the N = 4M 5-point Poisson matrix (a grid-line end in one tile of
eight) against a banded matrix with the same offsets and no line ends (all tiles uniform but the first and last eight)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd"))
import torch  # noqa: E402
from pytorch_sparse_solver import _hipk  # noqa: E402
from pytorch_sparse_solver.utils.matrix_utils import create_poisson_2d_csr  # noqa: E402

dev = torch.device("cuda", 0)
nx = 2000
n = nx * nx


def banded():
    offs = np.array([-nx, -1, 0, 1, nx])
    rows = np.repeat(np.arange(n), 5)
    cols = rows + np.tile(offs, n)
    vals = np.tile(np.array([-1.0, -1.0, 4.0, -1.0, -1.0]), n)
    keep = (cols >= 0) & (cols < n)
    rows, cols, vals = rows[keep], cols[keep], vals[keep]
    crow = np.zeros(n + 1, dtype=np.int64)
    np.add.at(crow, rows + 1, 1)
    return torch.from_numpy(np.cumsum(crow)).to(dev), torch.from_numpy(cols).to(dev), torch.from_numpy(vals).to(dev)


def time_modes(h):
    out = {}
    L = _hipk.lib()
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(n, dtype=torch.float64, device=dev, generator=g)
    y = torch.empty_like(x)
    G = int(L.hipk_chunk_count(n))
    p0 = torch.zeros(G, dtype=torch.float64, device=dev)
    p1 = torch.zeros(G, dtype=torch.float64, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    for mode in (0, 1, 2):
        for _ in range(20):
            L.hipk_spmv_ex(h._h, x.data_ptr(), y.data_ptr(), mode, x.data_ptr(), x.data_ptr(), p0.data_ptr(), p1.data_ptr(), None, 0, s)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(400):
            L.hipk_spmv_ex(h._h, x.data_ptr(), y.data_ptr(), mode, x.data_ptr(), x.data_ptr(), p0.data_ptr(), p1.data_ptr(), None, 0, s)
        e1.record()
        torch.cuda.synchronize()
        out[f"mode{mode}_us"] = round(e0.elapsed_time(e1) / 400 * 1e3, 2)
    out["kernel"] = h.last_spmv_kernel()
    return out


A = create_poisson_2d_csr(nx, nx, device=dev)
hp = _hipk.CsrHandle(A.crow_indices(), A.col_indices(), A.values(), A.shape)
print(json.dumps({"matrix": "poisson 2000x2000", **time_modes(hp)}))
crow, col, val = banded()
hb = _hipk.CsrHandle(crow, col, val, (n, n))
print(json.dumps({"matrix": "banded, no line ends", "path": hb.path(), **time_modes(hb)}))
