#!/usr/bin/env python3
"""Handle-creation cost (structure analysis + coded forms) at N = 4M: Poisson (pair-coded) and variable diffusion (offset-coded)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd"))
import torch
from pytorch_sparse_solver import _hipk
from pytorch_sparse_solver.utils.matrix_utils import create_poisson_2d_csr, create_variable_diffusion_2d_csr
for name, A in (("poisson", create_poisson_2d_csr(2000, 2000, device="cuda:0")),
                ("vardiff", create_variable_diffusion_2d_csr(2000, 2000, device="cuda:0"))):
    crow, col, val = A.crow_indices(), A.col_indices(), A.values()
    for rep in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        h = _hipk.CsrHandle(crow, col, val, A.shape)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
        print(f"{name} rep {rep}: create {dt:.2f} ms path {h.path()}", flush=True)
        h.close()
    os.environ["HIPK_SPMV_CODED"] = "0"
    torch.cuda.synchronize(); t0 = time.perf_counter()
    h = _hipk.CsrHandle(crow, col, val, A.shape); torch.cuda.synchronize()
    print(f"{name}: create without coded forms {(time.perf_counter() - t0) * 1e3:.2f} ms", flush=True)
    h.close(); del os.environ["HIPK_SPMV_CODED"]
