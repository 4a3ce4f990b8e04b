#!/usr/bin/env python3
"""Round-2 additions to tests/golden/, produced by RUNNING THE REFERENCE (imported from /root/reference/src; build
container only -- the reference never travels).  Data only: inputs and the reference's outputs.
Usage:  PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_r2.py

1. BASELINE config 4 at its quoted size: the lid-driven-cavity pressure systems of FVM steps 0..2 at the reference's
   DEFAULT nx = 100, Re = 400 (ldc_solver_common.py:32-36), recorded from the reference's own BaseLDCSolver, solved by
   the reference with exactly the call of ldc_solver_module_a.py:19-21 (`gmres(tol=1e-10, maxiter=1000, restart=30)`,
   and `bicgstab(tol=1e-10, maxiter=1000)`) -> tests/golden/ldc_nx100_step{0,1,2}.npz + ldc100_index.json.
2. The GMRES tolerances (TSL:735-753) of every GMRES fixture, for BOTH device branches -> tests/golden/gmres_tol.json:
   * `cpu`: captured from the reference itself (the `atol` / `ptol` arguments its `gmres` hands to
     `_gmres_solve_with_method`, recorded by wrapping that function while the reference runs);
   * `cuda`: this container has no GPU, so the reference cannot take its `device.type == 'cuda'` branch here.  The
     values are evaluated with the same torch expressions and the branch's constants (1e-12, eps*1000, TSL:737-740);
     the identical evaluation with the cpu constants is asserted equal, bit for bit, to what the reference captured.
   tests/test_oracle_golden.py pins the oracle's `gpu_tolerances` 0/1 branches to these numbers.
"""
import contextlib
import io
import json
import os
import sys
import warnings

import numpy as np
import torch

warnings.filterwarnings("ignore")
REF = "/root/reference"
sys.path.insert(0, os.path.join(REF, "src"))
sys.path.insert(0, os.path.join(REF, "FVM_example", "LDC_by_torchsp"))
sys.dont_write_bytecode = True

import pytorch_sparse_solver.module_a.torch_sparse_linalg as TSL  # noqa: E402
from pytorch_sparse_solver.module_a import bicgstab, gmres  # noqa: E402

OUT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden"))


class Counting:
    def __init__(self, A):
        self.A = A
        self.count = 0

    def __call__(self, v):
        self.count += 1
        return torch.matmul(self.A, v)


# ---- capture of the reference's effective tolerances (cpu branch, the real thing)
_captured = {}
_orig_solve = TSL._gmres_solve_with_method


def _recording_solve(A, b, x0, atol, ptol, *rest, **kw):
    _captured["atol_eff"] = float(atol)
    _captured["ptol"] = float(ptol)
    return _orig_solve(A, b, x0, atol, ptol, *rest, **kw)


TSL._gmres_solve_with_method = _recording_solve


def tolerances(b, tol, atol, cuda: bool):
    """atol_eff and ptol (M = identity) by the expressions of TSL:735-753 with the constants of the chosen branch."""
    size = b.numel()
    b_norm = torch.sqrt(torch.clamp(torch.vdot(b, b).real, min=0.0))
    if cuda:
        adaptive_tol = max(tol, 1e-12 * torch.sqrt(torch.tensor(size, dtype=torch.float64)))
        base_atol = torch.finfo(b.dtype).eps * 1000 * size
    else:
        adaptive_tol = max(tol, 1e-14 * torch.sqrt(torch.tensor(size, dtype=torch.float64)))
        base_atol = torch.finfo(b.dtype).eps * 100 * size
    atol_t = torch.maximum(torch.tensor(adaptive_tol) * b_norm,
                           torch.maximum(torch.tensor(atol), torch.tensor(base_atol)))
    ptol = b_norm * torch.minimum(torch.tensor(1.0), atol_t / b_norm)
    return float(atol_t), float(ptol)


def csr_of(d):
    n = int(d["n"])
    return torch.sparse_csr_tensor(torch.from_numpy(d["crow"]).long(), torch.from_numpy(d["col"]).long(),
                                   torch.from_numpy(d["val"]), size=(n, n))


def ldc_nx100(index):
    from ldc_solver_common import BaseLDCSolver

    class Recorder(BaseLDCSolver):
        def __init__(self, *a, **k):
            self.rhs = []
            super().__init__(*a, solver_label="golden", **k)

        def _solve_linear_system(self, prhs):
            self.rhs.append(prhs.clone())
            return gmres(self.A_csr, prhs, tol=1e-10, maxiter=1000, restart=30)   # ldc_solver_module_a.py:21 on the CSR form

    with contextlib.redirect_stdout(io.StringIO()):
        s = Recorder(device="cpu")                    # defaults: nx = 100, Re = 400
        for _ in range(3):
            s.step()
    assert s.nx == 100 and s.Re == 400.0
    A = s.A_csr
    crow, col, val = (A.crow_indices().numpy().astype(np.int32), A.col_indices().numpy().astype(np.int32),
                      A.values().numpy().astype(np.float64))
    for step, prhs in enumerate(s.rhs):
        name = f"ldc_nx100_step{step}"
        arrays = {}
        runs = [("gmres_batched", gmres, {"tol": 1e-10, "maxiter": 1000, "restart": 30}),
                ("gmres_incremental", gmres, {"tol": 1e-10, "maxiter": 1000, "restart": 30, "solve_method": "incremental"}),
                ("bicgstab", bicgstab, {"tol": 1e-10, "maxiter": 1000})]
        for tag, fn, kw in runs:
            op = Counting(A)
            x, info = fn(op, prhs, **kw)
            res = torch.norm(prhs - A @ x).item()
            arrays[f"{tag}_x"] = x.numpy()
            index.append({"case": name, "tag": tag, "solver": fn.__name__, "kwargs": kw, "has_x0": False, "info": int(info),
                          "matvecs": int(op.count), "residual_norm": res, "b_norm": torch.norm(prhs).item(),
                          "x_norm": torch.norm(x).item()})
            print(f"  {name} {tag:18s} info={info:2d} matvecs={op.count:5d} relres={res / torch.norm(prhs).item():.3e}")
        np.savez_compressed(os.path.join(OUT, name + ".npz"), crow=crow, col=col, val=val, b=prhs.numpy(), n=np.int64(A.shape[0]),
                            **arrays)


def gmres_tolerances():
    runs = json.load(open(os.path.join(OUT, "index.json")))["runs"]
    runs += json.load(open(os.path.join(OUT, "ldc100_index.json")))["runs"]
    out = []
    for r in runs:
        if r["solver"] != "gmres":
            continue
        d = np.load(os.path.join(OUT, r["case"] + ".npz"))
        A, b = csr_of(d), torch.from_numpy(d["b"])
        kw = dict(r["kwargs"])
        x0 = torch.from_numpy(d["x0"]) if r["has_x0"] else None
        _captured.clear()
        kw1 = dict(kw, maxiter=1)                       # the tolerances do not depend on how long the solve runs
        gmres(A, b, x0=x0, **kw1)
        tol, atol = kw.get("tol", 1e-5), kw.get("atol", 0.0)
        cpu = tolerances(b, tol, atol, cuda=False)
        assert cpu == (_captured["atol_eff"], _captured["ptol"]), (r["case"], r["tag"], cpu, _captured)
        cuda = tolerances(b, tol, atol, cuda=True)
        out.append({"case": r["case"], "tag": r["tag"], "tol": tol, "atol": atol,
                    "atol_eff_cpu": cpu[0], "ptol_cpu": cpu[1], "atol_eff_cuda": cuda[0], "ptol_cuda": cuda[1],
                    "cpu_values_captured_from_reference": True})
    with open(os.path.join(OUT, "gmres_tol.json"), "w") as f:
        json.dump({"generator": "oracle/gen_golden_r2.py", "torch": torch.__version__, "runs": out}, f, indent=1)
    print(f"wrote tolerances of {len(out)} GMRES runs")


def main():
    torch.set_num_threads(4)
    index = []
    ldc_nx100(index)
    with open(os.path.join(OUT, "ldc100_index.json"), "w") as f:
        json.dump({"generator": "oracle/gen_golden_r2.py", "torch": torch.__version__,
                   "reference": "Litianyu141/Pytorch-Sparse-Linalg-torch-amgx.cg.bicg.gmres @ /root/reference",
                   "note": "BASELINE config 4 at the reference's default size: BaseLDCSolver(nx=100, Re=400), FVM steps 0..2",
                   "runs": index}, f, indent=1)
    gmres_tolerances()


if __name__ == "__main__":
    main()
