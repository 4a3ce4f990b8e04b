"""The product's GENERIC path (callable A / CPU tensors / M / PyTrees) against the reference
fixtures, plus the API-surface behaviour of the reference's own tests
(test_module_a.py:93-315, test_unified.py:53-344, ci.yml:39-80) on CPU."""
import numpy as np
import pytest
import torch

from conftest import BICGSTAB_MATVEC_BAND, golden_index, golden_runs, load_case, run_id
from pytorch_sparse_solver import SparseSolver, get_available_backends, solve
from pytorch_sparse_solver import cg as top_cg
from pytorch_sparse_solver.module_a import (Partial, bicgstab, bicgstab_differentiable, cg, cg_differentiable,
                                            get_last_stats, gmres, gmres_differentiable, tree_map)

SOLVERS = {"cg": cg, "bicgstab": bicgstab, "gmres": gmres}


def csr_of(d):
    n = int(d["n"])
    return torch.sparse_csr_tensor(torch.from_numpy(d["crow"]).long(), torch.from_numpy(d["col"]).long(),
                                   torch.from_numpy(d["val"]), size=(n, n))


@pytest.mark.parametrize("r", golden_runs(), ids=run_id)
def test_generic_path_reproduces_reference(r):
    d = load_case(r["case"])
    kw = dict(r["kwargs"])
    if r["has_x0"]:
        kw["x0"] = torch.from_numpy(d["x0"])
    x, info = SOLVERS[r["solver"]](csr_of(d), torch.from_numpy(d["b"]), **kw)
    st = get_last_stats()
    x_ref = d[r["tag"] + "_x"]
    rel = np.linalg.norm(x.numpy() - x_ref) / max(np.linalg.norm(x_ref), 1e-300)
    assert info == r["info"] and st.matvecs == r["matvecs"] and rel < 1e-8
    assert isinstance(info, int) and x.dtype == torch.float64


def _spd(n, seed=42):
    g = torch.Generator().manual_seed(seed)
    G = torch.randn(n, n, dtype=torch.float64, generator=g)
    return G @ G.T + n * torch.eye(n, dtype=torch.float64), g


def test_matrix_free_and_preconditioned_cg():
    A, g = _spd(60)
    b = A @ torch.randn(60, dtype=torch.float64, generator=g)
    x_t, info_t = cg(A, b, tol=1e-10)
    x_f, info_f = cg(lambda v: A @ v, b, tol=1e-10)
    assert info_t == 0 and info_f == 0 and torch.allclose(x_t, x_f, rtol=1e-12, atol=1e-14)
    dinv = 1.0 / torch.diagonal(A)
    x_m, info_m = cg(A, b, tol=1e-10, M=lambda v: dinv * v)
    assert info_m == 0 and torch.norm(b - A @ x_m) / torch.norm(b) < 1e-9
    for fn in (bicgstab, gmres):
        x, info = fn(A, b, tol=1e-10, M=lambda v: dinv * v)
        assert info == 0 and torch.norm(b - A @ x) / torch.norm(b) < 1e-8


def test_pytree_operands():
    A, g = _spd(30)
    b = A @ torch.randn(30, dtype=torch.float64, generator=g)
    tree_b = {"u": b[:10].reshape(2, 5), "v": (b[10:18], [b[18:]])}

    def op(t):
        flat = torch.cat([t["u"].reshape(-1), t["v"][0], t["v"][1][0]])
        y = A @ flat
        return {"u": y[:10].reshape(2, 5), "v": (y[10:18], [y[18:]])}

    x_flat, _ = cg(A, b, tol=1e-10)
    for fn in (cg, bicgstab, gmres):
        x, info = fn(op, tree_b, tol=1e-10)
        assert info == 0 and x["u"].shape == (2, 5) and isinstance(x["v"], tuple) and isinstance(x["v"][1], list)
        got = torch.cat([x["u"].reshape(-1), x["v"][0], x["v"][1][0]])
        assert torch.allclose(got, x_flat, rtol=1e-7, atol=1e-9)
    assert tree_map(lambda a, b_: a + b_, {"k": 1, "j": [2, 3]}, {"k": 10, "j": [20, 30]}) == {"k": 11, "j": [22, 33]}
    assert Partial(lambda a, b_, c=0: a + b_ + c, 1, c=5)(2) == 8


def test_complex_hermitian_cg_and_gmres():
    g = torch.Generator().manual_seed(3)
    G = torch.randn(24, 24, dtype=torch.complex128, generator=g)
    A = G @ G.conj().T + 24 * torch.eye(24, dtype=torch.complex128)
    b = torch.randn(24, dtype=torch.complex128, generator=g)
    for fn in (cg, bicgstab, gmres):
        x, info = fn(A, b, tol=1e-10)
        assert x.dtype == torch.complex128 and info == 0
        assert torch.linalg.norm(b - A @ x) / torch.linalg.norm(b) < 1e-8


def test_fp32_rhs_is_promoted_and_errors_match_reference():
    A, g = _spd(20)
    b32 = torch.randn(20, generator=g)
    x, info = cg(A, b32, tol=1e-8)
    assert x.dtype == torch.float64 and info == 0                       # TSL:979-980
    with pytest.raises(ValueError, match="square matrix"):
        cg(torch.zeros(3, 4, dtype=torch.float64), torch.zeros(4, dtype=torch.float64))   # TSL:181-183
    with pytest.raises(TypeError, match="function or tensor"):
        cg("not an operator", b32)                                      # TSL:207-208
    with pytest.raises(ValueError, match="matching tree structure"):
        cg(lambda t: t, [b32, b32], x0=[b32])                           # TSL:996
    with pytest.raises(ValueError, match="matching shapes"):
        cg(A, b32, x0=torch.zeros(19))                                  # TSL:1000-1002
    with pytest.raises(ValueError, match="Unsupported solve_method"):
        gmres(A, b32, solve_method="qr")                                # TSL:760


def test_autograd_implicit_diff_matches_dense_solve():
    A, g = _spd(25)
    for fn, kw in ((cg, {}), (bicgstab, {}), (gmres, {"restart": 25})):
        b = torch.randn(25, dtype=torch.float64, generator=g).requires_grad_(True)
        x, info = fn(A, b, tol=1e-10, **kw)
        x.sum().backward()
        expect = torch.linalg.solve(A.T, torch.ones(25, dtype=torch.float64))
        assert info == 0 and torch.allclose(b.grad, expect, rtol=1e-6, atol=1e-9)
    for fn in (cg_differentiable, bicgstab_differentiable, gmres_differentiable):
        b = torch.randn(25, dtype=torch.float64, generator=g).requires_grad_(True)
        fn(A, b, tol=1e-10).sum().backward()
        assert torch.isfinite(b.grad).all() and b.grad.abs().sum() > 0   # test_gpu_validation.py:59-69
    with pytest.raises(ValueError, match="2D tensor"):
        cg_differentiable(lambda v: v, torch.ones(3, dtype=torch.float64))


def test_dispatcher_module_a_record_matches_reference():
    """solver.py:320-379 on the fixture captured from the reference dispatcher."""
    rec = golden_index()["dispatcher_poisson_nx16"]
    d = load_case("poisson_nx16")
    A, b = csr_of(d), torch.from_numpy(d["b"])
    for e in rec:
        x, res = SparseSolver().solve(A, b, method=e["method"], backend="module_a", tol=1e-6, **e["kwargs"])
        assert res.converged == e["converged"] and res.iterations is None and res.backend == "module_a"
        assert res.method == e["method"]
        assert abs(res.residual - e["residual"]) <= 1e-6 * max(e["residual"], 1e-12) + 1e-13
        assert abs(torch.norm(x).item() - e["x_norm"]) <= 1e-9 * e["x_norm"]


def test_dispatcher_surface():
    assert get_available_backends() == {"module_a": True, "module_b": False, "module_c": False}
    s = SparseSolver()
    assert s.available_backends == ["module_a"]
    A, g = _spd(100)                                    # ci.yml:49-77
    b = torch.randn(100, dtype=torch.float64, generator=g)
    for method, thr in (("cg", 1e-5), ("bicgstab", 1e-5), ("gmres", 1e-4)):
        x, r = solve(A, b, method=method, tol=1e-6)
        assert r.converged and r.residual < thr
        x, r = getattr(s, method)(A, b, tol=1e-6)
        assert r.residual <= 1e-4                       # test_unified.py:156-184
    x, r = top_cg(A, b)
    assert r.converged
    with pytest.raises(ValueError, match="not available"):
        s.solve(A, b, backend="module_b")               # solver.py:219-225
    with pytest.raises(ValueError, match="not available"):
        s.solve(A, b, backend="invalid_backend")        # test_unified.py:314-344
    with pytest.raises(ValueError, match="not available in Module A"):
        s.solve(A, b, method="amg", backend="module_a")
    with pytest.raises(ValueError):
        s.direct(A, b)
