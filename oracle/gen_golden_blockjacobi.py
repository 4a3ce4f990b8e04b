#!/usr/bin/env python3
"""tests/golden/bj_*.npz: block-Jacobi preconditioned runs of THE REFERENCE (imported from /root/reference/src, build
container only), through its own preconditioner hook `M` (TSL:849, 908, 922, 351) with
    M = lambda v: blockdiag(A)^-1 v        (batched matmul with the inverted diagonal blocks)
Data only: CSR inputs, b, the inverted blocks, and the reference's x / info / operator-application counts.
Usage:  PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_blockjacobi.py"""
import json
import os
import sys
import warnings

import numpy as np
import torch

warnings.filterwarnings("ignore")
sys.path.insert(0, "/root/reference/src")
sys.dont_write_bytecode = True
from pytorch_sparse_solver.module_a import bicgstab, cg, gmres  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd"))
OUT = os.path.normpath(os.path.join(HERE, "..", "tests", "golden"))


def load_builders():
    """Matrix builders of THIS package (inputs only; the solves below are the reference's)."""
    import importlib.util
    p = os.path.join(HERE, "..", "pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd", "pytorch_sparse_solver", "utils",
                     "matrix_utils.py")
    spec = importlib.util.spec_from_file_location("_mu", p)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


class Counting:
    def __init__(self, A):
        self.A, self.count = A, 0

    def __call__(self, v):
        self.count += 1
        return torch.matmul(self.A, v)


def blocks_inv(A_dense, bs):
    n = A_dense.shape[0]
    nb = (n + bs - 1) // bs
    B = torch.zeros(nb, bs, bs, dtype=torch.float64)
    for b in range(nb):
        lo, hi = b * bs, min(n, (b + 1) * bs)
        B[b, :hi - lo, :hi - lo] = A_dense[lo:hi, lo:hi]
        for j in range(hi - lo, bs):
            B[b, j, j] = 1.0
    return torch.linalg.inv(B)


def main():
    mu = load_builders()
    torch.set_num_threads(4)
    index = []
    cases = [("bj_varpoisson_nx32_bs4", mu.create_variable_diffusion_2d_csr(32, 32, contrast=2.0, seed=1), 4, True),
             ("bj_varpoisson_30x26_bs8", mu.create_variable_diffusion_2d_csr(30, 26, contrast=1.5, seed=2), 8, True),
             ("bj_convdiff_nx32_bs4", mu.create_convdiff_2d_csr(32, 32), 4, False),
             ("bj_convdiff_29x31_bs3", mu.create_convdiff_2d_csr(29, 31), 3, False)]
    for name, A, bs, spd in cases:
        n = A.shape[0]
        binv = blocks_inv(A.to_dense(), bs)
        nb = binv.shape[0]

        def M(v, binv=binv, nb=nb, bs=bs, n=n):
            vp = torch.zeros(nb * bs, dtype=v.dtype)
            vp[:n] = v
            return torch.bmm(binv, vp.view(nb, bs, 1)).view(-1)[:n]

        g = torch.Generator().manual_seed(n)
        b = A @ torch.randn(n, dtype=torch.float64, generator=g)
        runs = []
        if spd:
            runs.append(("cg", cg, {"tol": 1e-8}))
            runs.append(("cg_plain", cg, {"tol": 1e-8, "_noM": True}))
        runs.append(("bicgstab", bicgstab, {"tol": 1e-8}))
        runs.append(("gmres_batched", gmres, {"tol": 1e-8, "restart": 20}))
        runs.append(("gmres_incremental", gmres, {"tol": 1e-8, "restart": 20, "solve_method": "incremental"}))
        arrays = {}
        for tag, fn, kw in runs:
            kw = dict(kw)
            noM = kw.pop("_noM", False)
            op = Counting(A)
            x, info = fn(op, b, M=None if noM else M, **kw)
            res = torch.norm(b - A @ x).item()
            arrays[tag + "_x"] = x.numpy()
            index.append({"case": name, "tag": tag, "solver": fn.__name__, "kwargs": kw, "block_size": bs, "preconditioned": not noM,
                          "info": int(info), "matvecs": int(op.count), "residual_norm": res, "b_norm": torch.norm(b).item()})
            print(f"  {name:26s} {tag:18s} info={info:2d} matvecs={op.count:5d} relres={res / torch.norm(b).item():.3e}")
        np.savez_compressed(os.path.join(OUT, name + ".npz"), crow=A.crow_indices().numpy().astype(np.int32),
                            col=A.col_indices().numpy().astype(np.int32), val=A.values().numpy(), b=b.numpy(), n=np.int64(n),
                            binv=binv.numpy(), **arrays)
    with open(os.path.join(OUT, "bj_index.json"), "w") as f:
        json.dump({"generator": "oracle/gen_golden_blockjacobi.py", "torch": torch.__version__,
                   "reference": "Litianyu141/Pytorch-Sparse-Linalg-torch-amgx.cg.bicg.gmres @ /root/reference", "runs": index}, f, indent=1)
    print(f"wrote {len(index)} runs")


if __name__ == "__main__":
    main()
