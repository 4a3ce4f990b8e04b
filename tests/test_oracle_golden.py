"""The CPU oracle (oracle/krylov_oracle.c) against fixtures produced by the reference itself
(oracle/gen_golden.py): this is what pins the oracle (no GPU needed)."""
import numpy as np
import pytest

from conftest import BICGSTAB_MATVEC_BAND, gmres_tol_runs, golden_runs, ldc100_runs, load_case, run_id


@pytest.mark.parametrize("r", golden_runs() + ldc100_runs(), ids=run_id)
def test_oracle_reproduces_reference(oracle, r):
    d = load_case(r["case"])
    kw = dict(r["kwargs"])
    x0 = d["x0"] if r["has_x0"] else None
    res = getattr(oracle, r["solver"])(d["crow"], d["col"], d["val"], d["b"], x0=x0, **kw)
    x_ref = d[r["tag"] + "_x"]
    rel = np.linalg.norm(res.x - x_ref) / max(np.linalg.norm(x_ref), 1e-300)
    tol = kw.get("tol", 1e-5)
    if r["solver"] == "bicgstab":
        assert abs(res.matvecs - r["matvecs"]) <= max(2, BICGSTAB_MATVEC_BAND * r["matvecs"])
        # same verdict, or the recurrence/true-residual gap of BiCGStab: on the ill-conditioned
        # tridiagonal system at tol=1e-10 the recurrence converges but the true residual stalls near 1e-8
        assert res.info == r["info"] or res.residual_norm <= 1e-6 * res.b_norm
        assert rel < 1e-3
    else:
        assert res.info == r["info"]
        assert res.matvecs == r["matvecs"]          # same iteration / cycle count as the reference
        assert rel < 1e-8                             # same iterate up to summation-order rounding
        assert abs(res.residual_norm - r["residual_norm"]) <= 0.5 * max(res.residual_norm, r["residual_norm"]) \
            + 1e-12 * r["b_norm"]


def test_reduction_spec_geometry(oracle):
    assert oracle.chunk_geom(1) == (2048, 1)
    assert oracle.chunk_geom(4_000_000) == (2048, 1954)
    assert oracle.chunk_geom(2048 * 2048) == (2048, 2048)
    assert oracle.chunk_geom(2048 * 2048 + 1) == (4096, 1025)
    assert oracle.chunk_geom(64_000_000) == (32768, 1954)


def test_dot_matches_numpy_and_is_thread_independent(oracle):
    rng = np.random.default_rng(0)
    for n in (1, 2, 3, 511, 512, 513, 2048, 2049, 100_003):
        a, b = rng.standard_normal(n), rng.standard_normal(n)
        oracle.set_threads(1)
        d1 = oracle.dot(a, b)
        oracle.set_threads(4)
        d4 = oracle.dot(a, b)
        oracle.set_threads(1)
        assert d1 == d4
        assert abs(d1 - float(np.dot(a, b))) <= 1e-12 * max(1.0, np.abs(a * b).sum())
    parts = oracle.dot_parts(np.ones(5000), np.ones(5000))
    assert parts.tolist() == [2048.0, 2048.0, 904.0] and oracle.reduce_parts(parts) == 5000.0


def test_spmv_matches_scipy(oracle):
    sp = pytest.importorskip("scipy.sparse")
    rng = np.random.default_rng(1)
    A = sp.random(300, 300, density=0.2, random_state=2, format="csr")   # rows ~60 nnz: long-row tree path
    A.sort_indices()
    x = rng.standard_normal(300)
    y = oracle.spmv(A.indptr, A.indices, A.data, x)
    assert np.allclose(y, A @ x, rtol=1e-13, atol=1e-13)
    b = rng.standard_normal(300)
    assert np.array_equal(oracle.spmv(A.indptr, A.indices, A.data, x, bsub=b), b - y)


@pytest.mark.parametrize("case,solver,kw,limit", [
    ("poisson_nx32", "cg32", {"tol": 1e-4}, 2e-4),
    ("poisson_nx64", "cg32", {"tol": 1e-4}, 2e-4),
    ("convdiff_nx32", "bicgstab32", {"tol": 1e-4}, 2e-4),
    ("spd_n100", "cg32", {"tol": 1e-5}, 2e-5),
    ("ldc_nx32_step0", "gmres32", {"tol": 1e-4, "restart": 30, "maxiter": 1000}, 1e-3),
    ("ldc_nx16_step1", "gmres32", {"tol": 1e-4, "restart": 30, "maxiter": 1000, "solve_method": "incremental"}, 1e-3),
])
def test_fp32_oracle_against_fp64_reference_solution(oracle, case, solver, kw, limit):
    """fp32 storage is an extension (the reference raises on fp32 A): its oracle is pinned against the
    reference's fp64 solution of the same system -- converged verdict and x within fp32 accuracy.  fp32 true
    residuals stall near eps32 * ||A|| ||x|| (about 3e-5 ||b|| on the 32x32 Poisson system), so tol is 1e-4 here."""
    d = load_case(case)
    res = getattr(oracle, solver)(d["crow"], d["col"], d["val"], d["b"], **kw)
    tag = {"cg32": "cg", "bicgstab32": "bicgstab", "gmres32": "gmres_batched"}[solver]
    x_ref = d[tag + "_x"].astype(np.float64)
    # singular LDC systems: compare modulo the constant null-space component
    x = res.x.astype(np.float64)
    if case.startswith("ldc"):
        x, x_ref = x - x.mean(), x_ref - x_ref.mean()
    # `info` keeps the reference's rule (true residual <= tol ||b||, TSL:1013): in fp32 the recurrence can stop a
    # hair above it, so the check here is on the residual itself
    assert res.x.dtype == np.float32 and res.info in (0, -1)
    assert res.residual_norm <= limit * res.b_norm
    assert np.linalg.norm(x - x_ref) <= 5e-3 * np.linalg.norm(x_ref)


@pytest.mark.parametrize("t", gmres_tol_runs(), ids=lambda t: f"{t['case']}-{t['tag']}")
def test_gmres_tolerance_branches_pinned(oracle, t):
    """the oracle's `gpu_tolerances` 0 / 1 branches (TSL:735-748) against tests/golden/gmres_tol.json.
    The cpu values were CAPTURED from the reference while it ran (the tolerance its gmres hands to the restart loop);
    the cuda values are the same torch expressions with that branch's constants (the reference cannot take its
    `device.type == 'cuda'` branch in the GPU-less build container; oracle/gen_golden_r2.py asserts the cpu evaluation
    equal to the captured values).  `threshold` = 10 x atol_eff (TSL:769).  Exact where the absolute floor decides;
    within 4 ulp where tol * ||b|| decides (||b|| from the spec's chunked dot vs torch.vdot)."""
    d = load_case(t["case"])
    for branch, flag in (("cpu", False), ("cuda", True)):
        res = oracle.gmres(d["crow"], d["col"], d["val"], d["b"], tol=t["tol"], atol=t["atol"], restart=5, maxiter=0,
                           gpu_tolerances=flag)
        want = 10.0 * t[f"atol_eff_{branch}"]
        assert abs(res.threshold - want) <= 9e-16 * want, (branch, res.threshold, want)
    assert t["cpu_values_captured_from_reference"]


@pytest.mark.parametrize("step", [0, 1, 2])
def test_config4_fp32_oracle_against_reference_at_nx100(oracle, step):
    """BASELINE config 4 as quoted -- LDC nx = 100, gmres(restart=30), fp32 storage -- with the documented fp32
    tolerance 1e-5 (an fp32 true residual stalls near 5e-6 ||b|| on this system; the reference's 1e-10 is out of reach):
    oracle32 against the reference's fp64 solution of the same system (modulo the constant null-space component)."""
    d = load_case(f"ldc_nx100_step{step}")
    res = oracle.gmres32(d["crow"], d["col"], d["val"], d["b"].astype(np.float32), tol=1e-5, restart=30, maxiter=1000,
                         gpu_tolerances=True)
    x, x_ref = res.x.astype(np.float64), d["gmres_batched_x"]
    x, x_ref = x - x.mean(), x_ref - x_ref.mean()
    assert res.info == 0 and res.residual_norm <= 1e-5 * res.b_norm
    assert np.linalg.norm(x - x_ref) <= 2e-3 * np.linalg.norm(x_ref)
