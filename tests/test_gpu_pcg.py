"""GPU parity of the Jacobi-preconditioned CG fast path (hipk_pcg_solve) against the oracle -- bit for bit -- and
against the reference-generated fixtures (same stopping iteration, x to 1e-8)."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ALL = json.load(open(os.path.join(GOLD, "pcg_index.json")))["runs"]
RUNS = [r for r in ALL if r.get("solver", "cg") == "cg"]
GM = [r for r in ALL if r.get("solver") == "gmres"]
BI = [r for r in ALL if r.get("solver") == "bicgstab"]


def rid(r):
    return f"{r['case']}-{r['tag']}"


def dev_csr(d):
    n = int(d["n"])
    return torch.sparse_csr_tensor(torch.from_numpy(d["crow"]).long(), torch.from_numpy(d["col"]).long(),
                                   torch.from_numpy(d["val"]), size=(n, n)).to(DEV)


@pytest.mark.parametrize("r", RUNS, ids=rid)
def test_pcg_bit_exact_vs_oracle_and_reference_counts(hipk, oracle, r):
    from pytorch_sparse_solver.module_a import JacobiPreconditioner, cg, get_last_stats
    d = np.load(os.path.join(GOLD, r["case"] + ".npz"))
    A = dev_csr(d)
    M = JacobiPreconditioner(A)
    dinv = M.dinv.cpu().numpy()
    x0 = torch.from_numpy(d["x0"]).to(DEV) if r["has_x0"] else None
    x, info = cg(A, torch.from_numpy(d["b"]).to(DEV), x0=x0, M=M, **r["kwargs"])
    st = get_last_stats()
    assert st.method == "pcg_jacobi"                                   # the HIP path ran, not the generic one
    ref = oracle.pcg_jacobi(d["crow"], d["col"], d["val"], dinv, d["b"], x0=d["x0"] if r["has_x0"] else None,
                            **r["kwargs"])
    assert np.array_equal(x.cpu().numpy(), ref.x)
    assert (info, st.iterations, st.matvecs) == (ref.info, ref.iterations, ref.matvecs)
    assert st.residual_norm == ref.residual_norm and st.recurrence_rs == ref.recurrence_rs
    assert info == r["info"] and st.matvecs == r["matvecs"]            # the reference's own counts
    x_ref = d[r["tag"] + "_x"]
    assert np.linalg.norm(x.cpu().numpy() - x_ref) <= 1e-8 * np.linalg.norm(x_ref)


def test_pcg_large_variable_coefficient_problem(hipk, oracle):
    """N = 1M variable-coefficient diffusion: bit-exact vs the oracle, and far fewer iterations than plain CG."""
    from pytorch_sparse_solver.module_a import JacobiPreconditioner, cg, get_last_stats
    from pytorch_sparse_solver.utils.matrix_utils import create_variable_diffusion_2d_csr
    nx = 1000
    A = create_variable_diffusion_2d_csr(nx, nx, device=DEV)
    n = nx * nx
    b = torch.randn(n, dtype=torch.float64, device=DEV, generator=torch.Generator(device=DEV).manual_seed(3))
    M = JacobiPreconditioner(A)
    x, info = cg(A, b, tol=1e-6, maxiter=300, M=M)
    st = get_last_stats()
    crow, col, val = (t.cpu().numpy() for t in (A.crow_indices(), A.col_indices(), A.values()))
    oracle.set_threads(8)
    ref = oracle.pcg_jacobi(crow, col, val, M.dinv.cpu().numpy(), b.cpu().numpy(), tol=1e-6, maxiter=300)
    oracle.set_threads(1)
    assert np.array_equal(x.cpu().numpy(), ref.x) and (st.iterations, st.info) == (ref.iterations, ref.info)
    x2, info2 = cg(A, b, tol=1e-6, maxiter=300)
    st2 = get_last_stats()
    assert st.residual_norm < 0.05 * st2.residual_norm                 # after the same 300 iterations


def test_pcg_fp32_storage_and_coded_matrix(hipk, oracle):
    """Constant-coefficient Poisson takes the coded SpMV path; fp32 storage runs the fp32 kernels."""
    from pytorch_sparse_solver.module_a import JacobiPreconditioner, cg, get_last_stats
    from pytorch_sparse_solver.utils.matrix_utils import create_poisson_2d_csr
    A = create_poisson_2d_csr(300, 300, device=DEV)
    n = A.shape[0]
    b = torch.randn(n, dtype=torch.float64, device=DEV, generator=torch.Generator(device=DEV).manual_seed(4))
    crow, col, val = (t.cpu().numpy() for t in (A.crow_indices(), A.col_indices(), A.values()))
    M = JacobiPreconditioner(A)
    x, info = cg(A, b, tol=1e-8, M=M)
    st = get_last_stats()
    assert hipk.handle_for(A).path() == "coded"
    ref = oracle.pcg_jacobi(crow, col, val, M.dinv.cpu().numpy(), b.cpu().numpy(), tol=1e-8)
    assert np.array_equal(x.cpu().numpy(), ref.x) and (st.iterations, info) == (ref.iterations, ref.info) and info == 0
    A32 = torch.sparse_csr_tensor(A.crow_indices(), A.col_indices(), A.values().float(), size=A.shape)
    M32 = JacobiPreconditioner(A32)
    x32, info32 = cg(A32, b.float(), tol=1e-4, M=M32)
    st32 = get_last_stats()
    ref32 = oracle.pcg_jacobi32(crow, col, val, M32.dinv.cpu().numpy(), b.float().cpu().numpy(), tol=1e-4)
    assert x32.dtype == torch.float32 and np.array_equal(x32.cpu().numpy(), ref32.x)
    assert st32.iterations == ref32.iterations


def test_preconditioner_of_another_shape_is_rejected(hipk):
    from pytorch_sparse_solver.module_a import JacobiPreconditioner, cg
    from pytorch_sparse_solver.utils.matrix_utils import create_poisson_2d_csr
    A = create_poisson_2d_csr(20, 20, device=DEV)
    M = JacobiPreconditioner(create_poisson_2d_csr(10, 10, device=DEV))
    with pytest.raises(ValueError, match="preconditioner shape"):
        cg(A, torch.ones(400, dtype=torch.float64, device=DEV), M=M)


@pytest.mark.parametrize("r", GM, ids=rid)
def test_gmres_jacobi_bit_exact_vs_oracle_and_reference_counts(hipk, oracle, r):
    from pytorch_sparse_solver.module_a import JacobiPreconditioner, get_last_stats, gmres
    d = np.load(os.path.join(GOLD, r["case"] + ".npz"))
    A = dev_csr(d)
    M = JacobiPreconditioner(A)
    x0 = torch.from_numpy(d["x0"]).to(DEV) if r["has_x0"] else None
    x, info = gmres(A, torch.from_numpy(d["b"]).to(DEV), x0=x0, M=M, **r["kwargs"])
    st = get_last_stats()
    assert st.method == "pgmres_jacobi"
    ref = oracle.gmres_jacobi(d["crow"], d["col"], d["val"], M.dinv.cpu().numpy(), d["b"],
                              x0=d["x0"] if r["has_x0"] else None, gpu_tolerances=True, **r["kwargs"])
    assert np.array_equal(x.cpu().numpy(), ref.x)
    assert (info, st.iterations, st.matvecs) == (ref.info, ref.iterations, ref.matvecs)
    assert st.residual_norm == ref.residual_norm
    x_ref = d[r["tag"] + "_x"]
    if st.matvecs == r["matvecs"]:          # the reference ran with the CPU tolerance branch (TSL:737-744)
        assert np.linalg.norm(x.cpu().numpy() - x_ref) <= 1e-7 * np.linalg.norm(x_ref)


def test_gmres_jacobi_on_a_large_offset_coded_matrix(hipk, oracle):
    """N = 1M variable-coefficient diffusion (offset-coded SpMV with the row-scaling epilogue): bit-exact vs the oracle,
    and the preconditioned residual drops much faster than the unpreconditioned one."""
    from pytorch_sparse_solver.module_a import JacobiPreconditioner, get_last_stats, gmres
    from pytorch_sparse_solver.utils.matrix_utils import create_variable_diffusion_2d_csr
    nx = 1000
    A = create_variable_diffusion_2d_csr(nx, nx, device=DEV)
    assert hipk.handle_for(A).path() == "offset_coded"
    n = nx * nx
    b = torch.randn(n, dtype=torch.float64, device=DEV, generator=torch.Generator(device=DEV).manual_seed(5))
    M = JacobiPreconditioner(A)
    x, info = gmres(A, b, tol=1e-8, restart=20, maxiter=3, M=M, solve_method="incremental")
    st = get_last_stats()
    crow, col, val = (t.cpu().numpy() for t in (A.crow_indices(), A.col_indices(), A.values()))
    oracle.set_threads(8)
    ref = oracle.gmres_jacobi(crow, col, val, M.dinv.cpu().numpy(), b.cpu().numpy(), tol=1e-8, restart=20, maxiter=3,
                              solve_method="incremental", gpu_tolerances=True)
    oracle.set_threads(1)
    assert np.array_equal(x.cpu().numpy(), ref.x) and st.matvecs == ref.matvecs
    x2, _ = gmres(A, b, tol=1e-8, restart=20, maxiter=3, solve_method="incremental")
    r1 = torch.linalg.norm(b - torch.mv(A, x)) / torch.linalg.norm(b)
    r2 = torch.linalg.norm(b - torch.mv(A, x2)) / torch.linalg.norm(b)
    assert r1 < 0.5 * r2


@pytest.mark.parametrize("r", BI, ids=rid)
def test_bicgstab_jacobi_bit_exact_vs_oracle(hipk, oracle, r):
    from pytorch_sparse_solver.module_a import JacobiPreconditioner, bicgstab, get_last_stats
    d = np.load(os.path.join(GOLD, r["case"] + ".npz"))
    A = dev_csr(d)
    M = JacobiPreconditioner(A)
    x0 = torch.from_numpy(d["x0"]).to(DEV) if r["has_x0"] else None
    x, info = bicgstab(A, torch.from_numpy(d["b"]).to(DEV), x0=x0, M=M, **r["kwargs"])
    st = get_last_stats()
    assert st.method == "pbicgstab_jacobi"
    ref = oracle.bicgstab_jacobi(d["crow"], d["col"], d["val"], M.dinv.cpu().numpy(), d["b"],
                                 x0=d["x0"] if r["has_x0"] else None, **r["kwargs"])
    assert np.array_equal(x.cpu().numpy(), ref.x)
    assert (info, st.iterations, st.matvecs, st.breakdown) == (ref.info, ref.iterations, ref.matvecs, ref.breakdown)
    assert st.residual_norm == ref.residual_norm
    assert info == r["info"] and abs(st.matvecs - r["matvecs"]) <= max(2, 0.15 * r["matvecs"])


def test_bicgstab_jacobi_large_nonsymmetric(hipk, oracle):
    """N = 1M convection-diffusion with a row-scaled (badly balanced) operator: Jacobi restores convergence;
    bit-exact vs the oracle for a fixed number of steps."""
    from pytorch_sparse_solver.module_a import JacobiPreconditioner, bicgstab, get_last_stats
    from pytorch_sparse_solver.utils.matrix_utils import create_convdiff_2d_csr
    nx = 1000
    C = create_convdiff_2d_csr(nx, nx, device=DEV)
    n = nx * nx
    g = torch.Generator(device=DEV).manual_seed(9)
    scale = torch.exp(3.0 * torch.rand(n, dtype=torch.float64, device=DEV, generator=g))
    rows = torch.repeat_interleave(torch.arange(n, device=DEV), C.crow_indices()[1:] - C.crow_indices()[:-1])
    A = torch.sparse_csr_tensor(C.crow_indices(), C.col_indices(), C.values() * scale[rows], size=C.shape)
    b = torch.randn(n, dtype=torch.float64, device=DEV, generator=g)
    M = JacobiPreconditioner(A)
    x, info = bicgstab(A, b, tol=1e-10, maxiter=40, M=M)
    st = get_last_stats()
    crow, col, val = (t.cpu().numpy() for t in (A.crow_indices(), A.col_indices(), A.values()))
    oracle.set_threads(8)
    ref = oracle.bicgstab_jacobi(crow, col, val, M.dinv.cpu().numpy(), b.cpu().numpy(), tol=1e-10, maxiter=40)
    oracle.set_threads(1)
    assert np.array_equal(x.cpu().numpy(), ref.x) and (st.iterations, st.matvecs) == (ref.iterations, ref.matvecs)


# ---- cg() with an arbitrary callable M: fused kernels around the callable (_hipk.solve_cg_callable) ----
@pytest.mark.parametrize("r", RUNS, ids=rid)
def test_cg_with_a_callable_M_equals_the_jacobi_fast_path_and_the_reference(hipk, oracle, r):
    """M = (v -> dinv * v) as a plain Python callable takes the step-API path; it has to reproduce the oracle (and so
    hipk_pcg_solve) bit for bit, and the reference's own counts and x."""
    from pytorch_sparse_solver.module_a import JacobiPreconditioner, cg, get_last_stats
    d = np.load(os.path.join(GOLD, r["case"] + ".npz"))
    A = dev_csr(d)
    dinv_t = JacobiPreconditioner(A).dinv
    x0 = torch.from_numpy(d["x0"]).to(DEV) if r["has_x0"] else None
    calls = []

    def M(v):
        calls.append(1)
        return dinv_t * v
    x, info = cg(A, torch.from_numpy(d["b"]).to(DEV), x0=x0, M=M, **r["kwargs"])
    st = get_last_stats()
    assert st.method == "cg_callable_M" and len(calls) >= st.iterations + 2
    ref = oracle.pcg_jacobi(d["crow"], d["col"], d["val"], dinv_t.cpu().numpy(), d["b"],
                            x0=d["x0"] if r["has_x0"] else None, **r["kwargs"])
    assert np.array_equal(x.cpu().numpy(), ref.x)
    assert (info, st.iterations, st.matvecs) == (ref.info, ref.iterations, ref.matvecs)
    assert st.residual_norm == ref.residual_norm
    assert info == r["info"] and st.matvecs == r["matvecs"]
    x_ref = d[r["tag"] + "_x"]
    assert np.linalg.norm(x.cpu().numpy() - x_ref) <= 1e-8 * np.linalg.norm(x_ref)


def test_cg_with_matrix_M_and_generic_path_agree(hipk, monkeypatch):
    """A dense matrix as M (the reference accepts tensors, TSL:176-208) and a non-diagonal callable: the fused path
    and the generic torch-op path (HIPK_CG_CALLABLE_M=0) stop at the same iteration with the same x to rounding."""
    from pytorch_sparse_solver.module_a import cg, get_last_stats
    from pytorch_sparse_solver.utils.matrix_utils import create_variable_diffusion_2d_csr
    A = create_variable_diffusion_2d_csr(40, 30, device=DEV)
    n = A.shape[0]
    b = torch.randn(n, dtype=torch.float64, device=DEV, generator=torch.Generator(device=DEV).manual_seed(5))
    Ad = A.to_dense()
    Minv = torch.diag(1.0 / torch.diagonal(Ad)) + 1e-3 * torch.eye(n, dtype=torch.float64, device=DEV)   # SPD matrix M
    tri = torch.tril(Ad)

    def gauss_seidel_sym(v):                                         # symmetric Gauss-Seidel: (D+L)^-T D (D+L)^-1
        y = torch.linalg.solve_triangular(tri, v.unsqueeze(-1), upper=False)
        y = torch.diagonal(Ad).unsqueeze(-1) * y
        return torch.linalg.solve_triangular(tri.T, y, upper=True).squeeze(-1)
    for M in (Minv, gauss_seidel_sym):
        out = {}
        for flag in ("1", "0"):
            monkeypatch.setenv("HIPK_CG_CALLABLE_M", flag)
            x, info = cg(A, b, M=M, tol=1e-10)
            st = get_last_stats()
            out[flag] = (x.clone(), info, st.iterations, st.method)
        assert out["1"][3] == "cg_callable_M" and out["0"][3] != "cg_callable_M"
        assert out["1"][1] == out["0"][1] == 0 and abs(out["1"][2] - out["0"][2]) <= 4      # different summation orders
        assert torch.linalg.norm(out["1"][0] - out["0"][0]) <= 1e-8 * torch.linalg.norm(out["0"][0])
    monkeypatch.delenv("HIPK_CG_CALLABLE_M", raising=False)


@pytest.mark.parametrize("r", BI, ids=rid)
def test_bicgstab_with_a_callable_M_equals_the_jacobi_fast_path(hipk, oracle, r):
    """bicgstab(M=<python callable>) runs the device-resident loop with a callback where the reference applies M
    (hipk_pbicgstab_solve_cb): with M = (v -> dinv * v) the iterates are the oracle's bit for bit."""
    from pytorch_sparse_solver.module_a import JacobiPreconditioner, bicgstab, get_last_stats
    d = np.load(os.path.join(GOLD, r["case"] + ".npz"))
    A = dev_csr(d)
    dinv_t = JacobiPreconditioner(A).dinv
    x0 = torch.from_numpy(d["x0"]).to(DEV) if r["has_x0"] else None
    x, info = bicgstab(A, torch.from_numpy(d["b"]).to(DEV), x0=x0, M=lambda v: dinv_t * v, **r["kwargs"])
    st = get_last_stats()
    assert st.method == "bicgstab_callable_M"
    ref = oracle.bicgstab_jacobi(d["crow"], d["col"], d["val"], dinv_t.cpu().numpy(), d["b"],
                                 x0=d["x0"] if r["has_x0"] else None, **r["kwargs"])
    assert np.array_equal(x.cpu().numpy(), ref.x)
    assert (info, st.iterations, st.matvecs, st.breakdown) == (ref.info, ref.iterations, ref.matvecs, ref.breakdown)
    assert abs(st.residual_norm - ref.residual_norm) <= 1e-12 * max(ref.residual_norm, 1e-300)   # chunk dot vs tiled dot
    assert info == r["info"]


def test_callable_M_that_raises_propagates_and_leaves_the_handle_usable(hipk):
    from pytorch_sparse_solver.module_a import bicgstab, cg
    from pytorch_sparse_solver.utils.matrix_utils import create_poisson_2d_csr
    A = create_poisson_2d_csr(50, 50, device=DEV)
    b = torch.ones(2500, dtype=torch.float64, device=DEV)

    class Boom(Exception):
        pass
    calls = []

    def bad(v):
        calls.append(1)
        if len(calls) > 3:
            raise Boom("preconditioner failed")
        return v
    for f in (bicgstab, cg):
        calls.clear()
        with pytest.raises(Boom):
            f(A, b, M=bad, tol=1e-10)
        x, info = f(A, b, tol=1e-8)                                  # the handle (and its lock) are fine afterwards
        assert info == 0
    with pytest.raises(ValueError):
        bicgstab(A, b, M=lambda v: v[:10], tol=1e-8)


@pytest.mark.parametrize("r", GM, ids=rid)
def test_gmres_with_a_callable_M_matches_the_jacobi_fast_path(hipk, oracle, r):
    """gmres(M=<python callable>): the device-resident cycle with a callback after every SpMV (hipk_pgmres_solve_cb).
    ||M(.)||^2 is a chunked dot here and a tiled dot fused into the SpMV on the Jacobi path, so residual norms (and
    through beta = ||r|| the basis) agree to rounding, not bit for bit: same counts, x to 1e-9 of the oracle's."""
    from pytorch_sparse_solver.module_a import JacobiPreconditioner, get_last_stats, gmres
    d = np.load(os.path.join(GOLD, r["case"] + ".npz"))
    A = dev_csr(d)
    dinv_t = JacobiPreconditioner(A).dinv
    x0 = torch.from_numpy(d["x0"]).to(DEV) if r["has_x0"] else None
    x, info = gmres(A, torch.from_numpy(d["b"]).to(DEV), x0=x0, M=lambda v: dinv_t * v, **r["kwargs"])
    st = get_last_stats()
    assert st.method == "gmres_callable_M"
    ref = oracle.gmres_jacobi(d["crow"], d["col"], d["val"], dinv_t.cpu().numpy(), d["b"],
                              x0=d["x0"] if r["has_x0"] else None, gpu_tolerances=True, **r["kwargs"])
    assert (info, st.iterations, st.matvecs) == (ref.info, ref.iterations, ref.matvecs)
    assert np.linalg.norm(x.cpu().numpy() - ref.x) <= 1e-9 * np.linalg.norm(ref.x)
    assert abs(st.residual_norm - ref.residual_norm) <= 1e-6 * max(ref.residual_norm, 1e-30) + 1e-14 * st.b_norm


def test_inner_solve_as_preconditioner_needs_its_own_matrix_object(hipk):
    """M = a few CG sweeps with the same operator (inexact inner solve): with the SAME tensor the binding refuses the
    nested use of the handle with a clear error instead of dead-locking; with a clone it runs on the fused path."""
    from pytorch_sparse_solver import _hipk
    from pytorch_sparse_solver.module_a import cg, get_last_stats
    from pytorch_sparse_solver.utils.matrix_utils import create_poisson_2d_csr
    A = create_poisson_2d_csr(40, 40, device=DEV)
    b = torch.ones(1600, dtype=torch.float64, device=DEV)
    with pytest.raises(_hipk.HipkError, match="nested solve"):
        cg(A, b, M=lambda r: cg(A, r, tol=1e-2)[0], tol=1e-8)
    A2 = create_poisson_2d_csr(40, 40, device=DEV)                   # its own storage -> its own handle
    x, info = cg(A, b, M=lambda r: cg(A2, r, tol=1e-14, maxiter=200)[0], tol=1e-8)
    assert info == 0 and get_last_stats().method == "cg_callable_M" and get_last_stats().iterations <= 3


def test_callable_M_fp32_storage_equals_the_jacobi_paths(hipk):
    """fp32 matrix and vectors: the callable-M paths run in fp32 storage like the Jacobi ones (dots in fp64):
    cg / bicgstab bit-identical to them, gmres to rounding."""
    from pytorch_sparse_solver.module_a import JacobiPreconditioner, bicgstab, cg, get_last_stats, gmres
    from pytorch_sparse_solver.utils.matrix_utils import create_variable_diffusion_2d_csr
    A64 = create_variable_diffusion_2d_csr(70, 60, device=DEV)
    A = torch.sparse_csr_tensor(A64.crow_indices(), A64.col_indices(), A64.values().float(), size=A64.shape)
    n = A.shape[0]
    b = torch.randn(n, dtype=torch.float32, device=DEV, generator=torch.Generator(device=DEV).manual_seed(2))
    J = JacobiPreconditioner(A)
    dinv = J.dinv
    assert dinv.dtype == torch.float32
    for f, kw, exact in ((cg, dict(tol=1e-5), True), (bicgstab, dict(tol=1e-5), True),
                         (gmres, dict(tol=1e-5, restart=30, maxiter=40), False)):
        xj, ij = f(A, b, M=J, **kw)
        sj = get_last_stats()
        xc, ic = f(A, b, M=lambda v: dinv * v, **kw)
        sc = get_last_stats()
        assert sc.method.endswith("callable_M") and xc.dtype == torch.float32
        assert (ij, sj.iterations) == (ic, sc.iterations)
        if exact:
            assert torch.equal(xj, xc)
        else:
            assert torch.linalg.norm(xj.double() - xc.double()) <= 1e-5 * torch.linalg.norm(xj.double())


# ---- cg() with a matrix-free operator on device vectors (_hipk.solve_cg_stepwise without a handle) ----
from conftest import golden_runs, load_case, run_id  # noqa: E402


@pytest.mark.parametrize("r", golden_runs("cg"), ids=run_id)
def test_matrix_free_cg_reproduces_the_reference_fixtures(hipk, r):
    """A as a Python callable (here: the handle's SpMV) on CUDA vectors runs the fused vector kernels with the device
    stop word; held to the bar of the generic path: the reference's info, operator-application count, x to 1e-8."""
    from pytorch_sparse_solver.module_a import cg, get_last_stats
    d = load_case(r["case"])
    n = int(d["n"])
    A = torch.sparse_csr_tensor(torch.from_numpy(d["crow"]).long(), torch.from_numpy(d["col"]).long(),
                                torch.from_numpy(d["val"]), size=(n, n)).to(DEV)
    h = hipk.handle_for(A)
    kw = dict(r["kwargs"])
    if r["has_x0"]:
        kw["x0"] = torch.from_numpy(d["x0"]).to(DEV)
    x, info = cg(lambda v: hipk.spmv(h, v), torch.from_numpy(d["b"]).to(DEV), **kw)
    st = get_last_stats()
    assert st.method == "cg_matrix_free" and x.dtype == torch.float64
    x_ref = d[r["tag"] + "_x"]
    rel = np.linalg.norm(x.cpu().numpy() - x_ref) / max(np.linalg.norm(x_ref), 1e-300)
    assert info == r["info"] and st.matvecs == r["matvecs"] and rel < 1e-8


def test_matrix_free_cg_with_M_and_dense_operator(hipk):
    from pytorch_sparse_solver.module_a import cg, get_last_stats
    g = torch.Generator().manual_seed(42)
    G = torch.randn(300, 300, dtype=torch.float64, generator=g)
    A = (G @ G.T + 300 * torch.eye(300, dtype=torch.float64)).to(DEV)
    b = torch.randn(300, dtype=torch.float64, generator=g).to(DEV)
    dinv = 1.0 / torch.diagonal(A)
    x_h, info_h = cg(A, b, tol=1e-10)
    x_f, info_f = cg(lambda v: A @ v, b, tol=1e-10)
    assert get_last_stats().method == "cg_matrix_free"
    x_m, info_m = cg(lambda v: A @ v, b, tol=1e-10, M=lambda v: dinv * v)
    assert get_last_stats().method == "cg_matrix_free_callable_M"
    assert info_h == info_f == info_m == 0
    assert torch.linalg.norm(x_f - x_h) <= 1e-10 * torch.linalg.norm(x_h)
    assert torch.linalg.norm(x_m - x_h) <= 1e-9 * torch.linalg.norm(x_h)
    with pytest.raises(ValueError):
        cg(lambda v: (A @ v)[:10], b)


# ---- bicgstab() / gmres() (and cg()) with a matrix-free operator on the device-resident C loops (hipk_op_create) ----
from conftest import BICGSTAB_MATVEC_BAND  # noqa: E402


def _fixture_system(hipk, r):
    d = load_case(r["case"])
    n = int(d["n"])
    A = torch.sparse_csr_tensor(torch.from_numpy(d["crow"]).long(), torch.from_numpy(d["col"]).long(),
                                torch.from_numpy(d["val"]), size=(n, n)).to(DEV)
    kw = dict(r["kwargs"])
    if r["has_x0"]:
        kw["x0"] = torch.from_numpy(d["x0"]).to(DEV)
    return d, A, torch.from_numpy(d["b"]).to(DEV), kw


@pytest.mark.parametrize("r", golden_runs("bicgstab") + golden_runs("gmres"), ids=run_id)
def test_matrix_free_bicgstab_gmres_reproduce_the_reference_fixtures(hipk, r):
    """a callable `A` (TSL:176-208) keeps bicgstab() and gmres() on the device-resident loops -- no host
    synchronisation per iteration, one per GMRES cycle.  (a) With the handle's own SpMV as the callable the result is the MATRIX
    solve's, bit for bit (same fused-dot spec in the epilogue kernel); (b) with torch's CSR product (other summation order) it is
    held to the generic path's bar: the reference's info, its operator-application count (BiCGStab: the chaotic band), x to 1e-8."""
    from pytorch_sparse_solver.module_a import bicgstab, get_last_stats, gmres
    fn = {"bicgstab": bicgstab, "gmres": gmres}[r["solver"]]
    d, A, b, kw = _fixture_system(hipk, r)
    h = hipk.handle_for(A)
    x_mat, info_mat = fn(A, b, **kw)
    st_mat = get_last_stats()
    x_op, info_op = fn(lambda v: hipk.spmv(h, v), b, **kw)
    st = get_last_stats()
    assert st.method == f"{r['solver']}_matrix_free" and x_op.dtype == torch.float64
    assert info_op == info_mat and (st.iterations, st.matvecs) == (st_mat.iterations, st_mat.matvecs)
    assert torch.equal(x_op, x_mat)
    x_t, info_t = fn(lambda v: A @ v, b, **kw)
    st = get_last_stats()
    x_ref = d[r["tag"] + "_x"]
    rel = np.linalg.norm(x_t.cpu().numpy() - x_ref) / max(np.linalg.norm(x_ref), 1e-300)
    if r["solver"] == "bicgstab":
        assert abs(st.matvecs - r["matvecs"]) <= max(2, BICGSTAB_MATVEC_BAND * r["matvecs"])
        assert info_t == r["info"] or st.residual_norm <= 1e-6 * st.b_norm
        assert rel < 1e-3
    else:
        # the fixture ran on the reference's cpu tolerance branch: the device branch is looser or equal
        assert info_t == r["info"] and st.matvecs <= r["matvecs"]
        xg = x_t.cpu().numpy()
        if r["case"].startswith("ldc"):
            xg, x_ref = xg - xg.mean(), x_ref - x_ref.mean()
        assert np.linalg.norm(xg - x_ref) <= 1e-5 * np.linalg.norm(x_ref)


def test_matrix_free_solvers_with_preconditioners_and_errors(hipk):
    from pytorch_sparse_solver.module_a import JacobiPreconditioner, bicgstab, cg, get_last_stats, gmres
    from pytorch_sparse_solver.utils.matrix_utils import create_convdiff_2d_csr
    A = create_convdiff_2d_csr(70, 60, device=DEV)
    n = A.shape[0]
    h = hipk.handle_for(A)
    b = torch.randn(n, dtype=torch.float64, device=DEV, generator=torch.Generator(device=DEV).manual_seed(3))
    op = lambda v: hipk.spmv(h, v)   # noqa: E731
    J = JacobiPreconditioner(A)
    for fn, kw in ((bicgstab, dict(tol=1e-9)), (gmres, dict(tol=1e-9, restart=25)), (gmres, dict(tol=1e-9, restart=40, solve_method="incremental"))):
        x_mat, info_mat = fn(A, b, M=J, **kw)                    # device-resident Jacobi, matrix operand
        x_op, info_op = fn(op, b, M=J, **kw)                     # the same loops around the callable
        assert get_last_stats().method.endswith("_matrix_free_jacobi")
        assert info_op == info_mat == 0 and torch.equal(x_op, x_mat)
        x_cb, info_cb = fn(op, b, M=lambda v: J.dinv * v, **kw)  # callable operator AND callable preconditioner
        assert get_last_stats().method.endswith("_matrix_free_callable_M")
        if fn is bicgstab:
            assert info_cb == 0 and torch.equal(x_cb, x_mat)     # M = diag(dinv): the Jacobi form's bits (include/hipk.h)
        else:   # the callback form takes ||M A v||^2 as a plain dot, the Jacobi form as the SpMV epilogue's tiled dot
            assert info_cb == 0 and torch.linalg.norm(x_cb - x_mat) <= 1e-8 * torch.linalg.norm(x_mat)
    with pytest.raises(ValueError):
        bicgstab(lambda v: (A @ v)[:10], b)                      # an exception inside the callback surfaces as itself
    with pytest.raises(ValueError):
        gmres(op, b, solve_method="nope")
    # many short solves: handles are created and destroyed per solve
    for _ in range(20):
        x, info = bicgstab(op, b, tol=1e-3)
    assert type(get_last_stats()).__name__ == "SolveStats"
