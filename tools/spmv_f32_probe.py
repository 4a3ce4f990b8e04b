#!/usr/bin/env python3
"""Is the coded SpMV bound by bytes or by memory instructions?  Same matrix in fp64 and fp32 storage."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd"))
import torch
from pytorch_sparse_solver import _hipk
from pytorch_sparse_solver.utils.matrix_utils import create_poisson_2d_csr
dev = torch.device("cuda", 0)
nx = 2000
A = create_poisson_2d_csr(nx, nx, device=dev)
for dtype in (torch.float64, torch.float32):
    h = _hipk.CsrHandle(A.crow_indices(), A.col_indices(), A.values().to(dtype), A.shape)
    x = torch.randn(nx * nx, dtype=dtype, device=dev)
    y = torch.empty_like(x)
    for plain in (False, True):
        h.set_path(plain_only=plain)
        for _ in range(20):
            _hipk.spmv(h, x, out=y)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            _hipk.spmv(h, x, out=y)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 200 * 1e3
        print(json.dumps({"dtype": str(dtype), "path": h.path(), "us": us, "format_MB": h.format_bytes() / 1e6,
                          "format_GBps": h.format_bytes() / us / 1e3}), flush=True)
