"""Block-Jacobi preconditioning: `BlockJacobiPreconditioner` is a callable for the
reference's `M` hook (TSL:849, 908, 922, 351) whose apply is a device kernel (hipk_block_jacobi_apply).
Fixtures: tests/golden/bj_*.npz -- the REFERENCE run with `M = lambda v: blockdiag(A)^-1 v` (oracle/gen_golden_blockjacobi.py)."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import BICGSTAB_MATVEC_BAND, GOLDEN, load_case


def _runs():
    with open(os.path.join(GOLDEN, "bj_index.json")) as f:
        return json.load(f)["runs"]


def _csr(d, device="cpu"):
    n = int(d["n"])
    return torch.sparse_csr_tensor(torch.from_numpy(d["crow"]).long(), torch.from_numpy(d["col"]).long(),
                                   torch.from_numpy(d["val"]), size=(n, n)).to(device)


RID = lambda r: f"{r['case']}-{r['tag']}"   # noqa: E731


@pytest.mark.parametrize("r", [r for r in _runs() if r["solver"] == "cg" and r["preconditioned"]], ids=RID)
def test_oracle_block_jacobi_cg_reproduces_reference(oracle, r):
    d = load_case(r["case"])
    res = oracle.pcg_blockjacobi(d["crow"], d["col"], d["val"], d["binv"], d["b"], **r["kwargs"])
    x_ref = d[r["tag"] + "_x"]
    assert res.info == r["info"] == 0 and res.matvecs == r["matvecs"]
    assert np.linalg.norm(res.x - x_ref) <= 1e-8 * np.linalg.norm(x_ref)


@pytest.mark.parametrize("case", sorted({r["case"] for r in _runs()}))
def test_preconditioner_blocks_and_cpu_apply(oracle, case):
    """The class inverts the same diagonal blocks as the fixture's generator; its CPU apply equals the oracle's to rounding."""
    from pytorch_sparse_solver.module_a import BlockJacobiPreconditioner
    d = load_case(case)
    bs = int(d["binv"].shape[1])
    M = BlockJacobiPreconditioner(_csr(d), block_size=bs)
    assert M.binv.shape == d["binv"].shape and torch.allclose(M.binv, torch.from_numpy(d["binv"]), rtol=1e-12, atol=1e-14)
    v = np.random.default_rng(0).standard_normal(int(d["n"]))
    z = M(torch.from_numpy(v)).numpy()
    assert np.allclose(z, oracle.block_jacobi_apply(M.binv.numpy(), v), rtol=1e-13, atol=1e-15)
    with pytest.raises(ValueError):
        BlockJacobiPreconditioner(_csr(d), block_size=33)
    with pytest.raises(ValueError):
        M(torch.zeros(3, dtype=torch.float64))


@pytest.mark.parametrize("r", [r for r in _runs() if r["preconditioned"]], ids=RID)
def test_generic_path_with_block_jacobi_matches_reference(r):
    """CPU tensors: this package's generic torch-op path with the class as `M` against the reference's own runs."""
    from pytorch_sparse_solver.module_a import BlockJacobiPreconditioner, bicgstab, cg, get_last_stats, gmres
    d = load_case(r["case"])
    A = _csr(d)
    M = BlockJacobiPreconditioner(A, block_size=r["block_size"])
    x, info = {"cg": cg, "bicgstab": bicgstab, "gmres": gmres}[r["solver"]](A, torch.from_numpy(d["b"]), M=M, **r["kwargs"])
    st = get_last_stats()
    x_ref = d[r["tag"] + "_x"]
    assert info == r["info"]
    if r["solver"] == "bicgstab":
        assert abs(st.matvecs - r["matvecs"]) <= max(2, BICGSTAB_MATVEC_BAND * r["matvecs"])
        assert np.linalg.norm(x.numpy() - x_ref) <= 1e-5 * np.linalg.norm(x_ref)
    else:
        assert st.matvecs == r["matvecs"]
        assert np.linalg.norm(x.numpy() - x_ref) <= 1e-8 * np.linalg.norm(x_ref)


# ------------------------------------------------------------------------------------------------ GPU
DEV = "cuda:0"


@pytest.mark.gpu
@pytest.mark.parametrize("n,bs,dt", [(1024, 4, torch.float64), (1000, 8, torch.float64), (899, 3, torch.float64),
                                     (4097, 16, torch.float64), (77, 32, torch.float64), (5, 7, torch.float64),
                                     (1024, 4, torch.float32), (901, 5, torch.float32)])
def test_block_jacobi_kernel_bit_exact(hipk, oracle, n, bs, dt):
    rng = np.random.default_rng(n + bs)
    nb = (n + bs - 1) // bs
    binv = rng.standard_normal((nb, bs, bs))
    v = rng.standard_normal(n)
    if dt == torch.float64:
        ref = oracle.block_jacobi_apply(binv, v)
    else:
        import ctypes
        b32, v32, out = binv.astype(np.float32), v.astype(np.float32), np.empty(n, np.float32)
        fp = ctypes.POINTER(ctypes.c_float)
        oracle.lib().orc32_block_jacobi_apply.argtypes = [ctypes.c_int64, ctypes.c_int, fp, fp, fp]
        oracle.lib().orc32_block_jacobi_apply(n, bs, b32.ctypes.data_as(fp), v32.ctypes.data_as(fp), out.ctypes.data_as(fp))
        ref = out
    z = hipk.block_jacobi_apply(torch.from_numpy(binv).to(dt).to(DEV).contiguous(), bs, torch.from_numpy(v).to(dt).to(DEV))
    assert np.array_equal(z.cpu().numpy(), ref)


@pytest.mark.gpu
@pytest.mark.parametrize("r", [r for r in _runs() if r["preconditioned"]], ids=RID)
def test_gpu_solvers_with_block_jacobi(hipk, oracle, r):
    """The fused kernels with the block-Jacobi kernel between them (cg: step API; bicgstab / gmres: the C loops' callback).
    CG: bit for bit the oracle (which reproduces the reference's count and x); BiCGStab / GMRES: the reference's verdict,
    its operator-application count (chaotic band for BiCGStab) and x."""
    from pytorch_sparse_solver.module_a import BlockJacobiPreconditioner, bicgstab, cg, get_last_stats, gmres
    d = load_case(r["case"])
    A = _csr(d, DEV)
    M = BlockJacobiPreconditioner(A, block_size=r["block_size"])
    assert M.binv.is_cuda
    x, info = {"cg": cg, "bicgstab": bicgstab, "gmres": gmres}[r["solver"]](A, torch.from_numpy(d["b"]).to(DEV), M=M, **r["kwargs"])
    st = get_last_stats()
    assert type(st).__name__ == "SolveStats" and "callable_M" in st.method        # the HIP path, not the generic one
    x_ref = d[r["tag"] + "_x"]
    xs = x.cpu().numpy()
    assert info == r["info"]
    if r["solver"] == "cg":
        ref = oracle.pcg_blockjacobi(d["crow"], d["col"], d["val"], M.binv.cpu().numpy(), d["b"], **r["kwargs"])
        assert st.matvecs == ref.matvecs == r["matvecs"] and np.array_equal(xs, ref.x)
    elif r["solver"] == "bicgstab":
        assert abs(st.matvecs - r["matvecs"]) <= max(2, BICGSTAB_MATVEC_BAND * r["matvecs"])
        assert np.linalg.norm(xs - x_ref) <= 1e-5 * np.linalg.norm(x_ref)
    else:
        assert st.matvecs <= r["matvecs"]                # GPU tolerance branch (TSL:737-740)
        assert np.linalg.norm(xs - x_ref) <= 1e-6 * np.linalg.norm(x_ref)
