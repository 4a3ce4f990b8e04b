"""GPU tests of the public API surface around the fast path: reference test recipes on the device
(test_module_a.py:93-315, test_gpu_validation.py:128-217, test_unified.py:90-228), generic path on CUDA tensors,
autograd through the fast path, handle cache reuse (LDC caller pattern), C-ABI error paths."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _spd(n, seed=42):
    g = torch.Generator().manual_seed(seed)
    G = torch.randn(n, n, dtype=torch.float64, generator=g)
    return (G @ G.T + n * torch.eye(n, dtype=torch.float64)).to(DEV), g


def test_reference_recipes_dense_inputs_on_gpu(hipk):
    """The reference feeds DENSE matrices: they are converted to CSR once and take the HIP path."""
    from pytorch_sparse_solver.module_a import bicgstab, cg, get_last_stats, gmres
    n = 100
    A = (2 * torch.eye(n, dtype=torch.float64) - torch.diag(torch.ones(n - 1, dtype=torch.float64), 1)
         - torch.diag(torch.ones(n - 1, dtype=torch.float64), -1)).to(DEV)
    g = torch.Generator().manual_seed(0)
    b = A @ torch.randn(n, dtype=torch.float64, generator=g).to(DEV)
    x, info = cg(A, b, tol=1e-10, maxiter=1000)                          # test_module_a.py:93-124
    assert info == 0 and (torch.norm(b - A @ x) / torch.norm(b)).item() < 1e-6
    assert type(get_last_stats()).__name__ == "SolveStats"                # the HIP path ran, not the generic one
    B = A + 0.1 * torch.randn(n, n, dtype=torch.float64, generator=g).to(DEV) + 5 * torch.eye(n, dtype=torch.float64).to(DEV)
    b2 = B @ torch.randn(n, dtype=torch.float64, generator=g).to(DEV)
    x, info = bicgstab(B, b2, tol=1e-10, maxiter=1000)                   # :126-161
    assert info == 0 and (torch.norm(b2 - B @ x) / torch.norm(b2)).item() < 1e-5
    C = torch.randn(n, n, dtype=torch.float64, generator=g).to(DEV) + 10 * torch.eye(n, dtype=torch.float64).to(DEV)
    b3 = C @ torch.randn(n, dtype=torch.float64, generator=g).to(DEV)
    for m in ("batched", "incremental"):
        x, info = gmres(C, b3, tol=1e-10, maxiter=1000, restart=30, solve_method=m)   # :163-195, :273-315
        assert info == 0 and (torch.norm(b3 - C @ x) / torch.norm(b3)).item() < 1e-5
    # COO input and fp32 right-hand side promotion
    x, info = cg(A.to_sparse_coo(), b.float(), tol=1e-6)
    assert info == 0 and x.dtype == torch.float64


def test_generic_path_on_cuda_tensors_matches_fast_path(hipk, monkeypatch):
    from pytorch_sparse_solver.module_a import cg, get_last_stats
    A, g = _spd(80)
    b = torch.randn(80, dtype=torch.float64, generator=g).to(DEV)
    x_fast, info_fast = cg(A, b, tol=1e-10)
    it_fast = get_last_stats().iterations
    dinv = 1.0 / torch.diagonal(A)
    # callable operator / callable preconditioner: by default between the fused kernels (step API) ...
    x_mf, info_mf = cg(lambda v: A @ v, b, tol=1e-10)
    assert get_last_stats().method == "cg_matrix_free" and get_last_stats().iterations == it_fast
    assert info_mf == 0 and torch.allclose(x_fast, x_mf, rtol=1e-10, atol=1e-12)
    # ... and with the switches off on the generic path: the reference's algorithm in torch ops on the GPU
    monkeypatch.setenv("HIPK_CG_MATRIX_FREE", "0")
    monkeypatch.setenv("HIPK_CG_CALLABLE_M", "0")
    x_gen, info_gen = cg(lambda v: A @ v, b, tol=1e-10)
    assert type(get_last_stats()).__name__ == "_GenericStats" and get_last_stats().iterations == it_fast
    assert info_fast == info_gen == 0 and torch.allclose(x_fast, x_gen, rtol=1e-10, atol=1e-12)
    x_m, info_m = cg(A, b, tol=1e-10, M=lambda v: dinv * v)
    assert type(get_last_stats()).__name__ == "_GenericStats"
    assert info_m == 0 and torch.allclose(x_m, x_fast, rtol=1e-7, atol=1e-9)


def test_autograd_on_the_fast_path(hipk):
    """test_gpu_validation.py:166-215: gradients exist, are finite and non-zero; here also checked against A^-T 1."""
    from pytorch_sparse_solver.module_a import (bicgstab, bicgstab_differentiable, cg, cg_differentiable, gmres,
                                                gmres_differentiable)
    A, g = _spd(96)
    expect = torch.linalg.solve(A.T, torch.ones(96, dtype=torch.float64, device=DEV))
    for fn, kw in ((cg, {}), (bicgstab, {}), (gmres, {"restart": 30})):
        b = torch.randn(96, dtype=torch.float64, generator=g).to(DEV).requires_grad_(True)
        x, info = fn(A.to_sparse_csr(), b, tol=1e-10, **kw)
        x.sum().backward()
        assert info == 0 and torch.allclose(b.grad, expect, rtol=1e-6, atol=1e-9)
    for fn in (cg_differentiable, bicgstab_differentiable, gmres_differentiable):
        b = torch.randn(96, dtype=torch.float64, generator=g).to(DEV).requires_grad_(True)
        fn(A, b, tol=1e-10).sum().backward()
        assert torch.isfinite(b.grad).all() and b.grad.abs().sum() > 0


def _convdiff(nx, device=DEV):
    """Nonsymmetric convection-diffusion matrix (BASELINE config 3 recipe: gamma 0.5, delta 0.25) as torch CSR."""
    from pytorch_sparse_solver.utils.matrix_utils import create_convdiff_2d_csr
    return create_convdiff_2d_csr(nx, nx, device=device)


@pytest.mark.parametrize("layout", ["csr", "dense"])
def test_autograd_nonsymmetric_adjoint_uses_the_transpose(hipk, layout):
    """ADVICE r1 (high) / VERDICT 3b: the adjoint solve of the implicit-diff backward (TSL:1237-1248) must run with
    A^T.  A dense `A` and its view `A.T` share storage and version: the handle cache must tell them apart.
    Recipe of test_gpu_validation.py:166-215 on a NONSYMMETRIC matrix; expected b.grad = A^-T g (torch.linalg.solve)."""
    from pytorch_sparse_solver.module_a import (bicgstab, bicgstab_differentiable, get_last_stats, gmres,
                                                gmres_differentiable)
    hipk.clear_cache()
    nx = 32
    A_csr = _convdiff(nx)
    A_dense = A_csr.to_dense()
    assert not torch.allclose(A_dense, A_dense.T)
    A = A_csr if layout == "csr" else A_dense
    n = nx * nx
    g = torch.Generator().manual_seed(3)
    gvec = torch.randn(n, dtype=torch.float64, generator=g).to(DEV)      # d(loss)/dx: loss = <gvec, x>
    expect = torch.linalg.solve(A_dense.T, gvec)
    wrong = torch.linalg.solve(A_dense, gvec)                            # what a solve with A instead of A^T would give
    assert (expect - wrong).norm() / expect.norm() > 1e-2
    for fn, kw in ((bicgstab, {}), (gmres, {"restart": 30})):
        b = torch.randn(n, dtype=torch.float64, generator=g).to(DEV).requires_grad_(True)
        x, info = fn(A, b, tol=1e-9, **kw)
        assert type(get_last_stats()).__name__ == "SolveStats"            # forward on the HIP path
        (x * gvec).sum().backward()
        assert type(get_last_stats()).__name__ == "SolveStats"            # adjoint solve on the HIP path as well
        assert info == 0
        assert (b.grad - expect).norm() / expect.norm() < 1e-6, (fn.__name__, (b.grad - expect).norm() / expect.norm())
    for fn, kw in ((bicgstab_differentiable, {}), (gmres_differentiable, {"restart": 30})):
        b = torch.randn(n, dtype=torch.float64, generator=g).to(DEV).requires_grad_(True)
        (fn(A, b, tol=1e-9, **kw) * gvec).sum().backward()
        assert (b.grad - expect).norm() / expect.norm() < 1e-6, (fn.__name__, (b.grad - expect).norm() / expect.norm())


def test_forward_solve_with_a_transposed_view(hipk):
    """solve(A, b) followed by solve(A.T, b) must give two different, correct answers (dense view and CSR .t())."""
    from pytorch_sparse_solver.module_a import bicgstab
    hipk.clear_cache()
    A_csr = _convdiff(24)
    A = A_csr.to_dense()
    g = torch.Generator().manual_seed(5)
    b = torch.randn(A.shape[0], dtype=torch.float64, generator=g).to(DEV)
    for M, Mt in ((A, A.T), (A_csr, A_csr.t())):
        x1, i1 = bicgstab(M, b, tol=1e-9)
        x2, i2 = bicgstab(Mt, b, tol=1e-9)
        assert i1 == 0 and i2 == 0
        r1 = torch.linalg.solve(A, b)
        assert (x1 - r1).norm() / r1.norm() < 1e-6
        r2 = torch.linalg.solve(A.T, b)
        assert (x2 - r2).norm() / r2.norm() < 1e-6
        assert (x1 - x2).norm() / x1.norm() > 1e-3


@pytest.mark.parametrize("kind", ["convdiff", "random", "dense", "empty_rows", "fp32"])
def test_native_transpose_handle_bit_exact(hipk, oracle, kind):
    """A^T built on the device from the handle's own arrays (hipk_csr_transpose).  Its SpMV must equal
    the oracle SpMV of the explicit, column-sorted transpose BIT FOR BIT (so the rows of A^T are sorted like torch's CSR)."""
    import scipy.sparse as sp
    rng = np.random.default_rng(11)
    if kind == "convdiff":
        A = _convdiff(40).cpu()
        M = sp.csr_matrix((A.values().numpy(), A.col_indices().numpy(), A.crow_indices().numpy()), shape=tuple(A.shape))
    elif kind == "dense":
        M = sp.csr_matrix(rng.standard_normal((300, 300)))
    elif kind == "empty_rows":
        M = sp.random(700, 700, density=0.004, random_state=3, format="csr")     # many empty rows and columns
    else:
        M = sp.random(3000, 3000, density=0.003, random_state=5, format="csr") + sp.eye(3000, format="csr") * 3.0
    M = M.tocsr()
    M.sort_indices()
    dt = np.float32 if kind == "fp32" else np.float64
    M = M.astype(dt)
    A = torch.sparse_csr_tensor(torch.from_numpy(M.indptr.astype(np.int64)), torch.from_numpy(M.indices.astype(np.int64)),
                                torch.from_numpy(M.data), size=M.shape).to(DEV)
    hipk.clear_cache()
    h = hipk.handle_for(A)
    ht = h.transposed()
    assert ht is h.transposed() and ht.shape == (M.shape[1], M.shape[0]) and ht.nnz == M.nnz
    Mt = M.T.tocsr()
    Mt.sort_indices()
    assert np.array_equal(ht.crow.cpu().numpy(), Mt.indptr) and np.array_equal(ht.col.cpu().numpy(), Mt.indices)
    assert np.array_equal(ht.val.cpu().numpy(), Mt.data)
    x = rng.standard_normal(M.shape[0]).astype(dt)
    y = hipk.spmv(ht, torch.from_numpy(x).to(DEV)).cpu().numpy()
    spmv = oracle.spmv32 if kind == "fp32" else oracle.spmv
    assert np.array_equal(y, spmv(Mt.indptr.astype(np.int32), Mt.indices.astype(np.int32), Mt.data, x))
    # the views torch hands to the adjoint solve resolve to the same cached transposed handle (no re-conversion)
    assert hipk.handle_for(A.t()) is ht
    Ad = A.to_dense()
    assert hipk.handle_for(Ad.T) is hipk.handle_for(Ad).transposed()


def test_dispatcher_on_gpu(hipk):
    from pytorch_sparse_solver import SparseSolver, solve
    A, g = _spd(100)
    b = torch.randn(100, dtype=torch.float64, generator=g).to(DEV)
    s = SparseSolver()
    x, r = s.solve(A, b, method="cg", backend="module_a", tol=1e-8)      # test_unified.py:90-127
    assert r.converged and r.residual < 1e-5 and r.backend == "module_a" and r.iterations is None
    for m in ("cg", "bicgstab", "gmres"):
        x, r = solve(A.to_sparse_csr(), b, method=m, tol=1e-6)
        assert r.converged and r.residual <= 1e-4


def test_handle_cache_reuse_for_repeated_solves(hipk):
    """The LDC stepper pattern (ldc_solver_common.py:185-201): one matrix, many right-hand sides."""
    from pytorch_sparse_solver.module_a import bicgstab
    from pytorch_sparse_solver.utils.matrix_utils import create_ldc_pressure_csr
    hipk.clear_cache()
    A = create_ldc_pressure_csr(48, device=DEV)
    g = torch.Generator().manual_seed(1)
    handles = set()
    for _ in range(4):
        b = torch.randn(48 * 48, dtype=torch.float64, generator=g).to(DEV)
        b -= b.mean()
        x, info = bicgstab(A, b, tol=1e-10, maxiter=1000)
        assert info == 0
        handles.add(id(hipk.handle_for(A)))
    assert len(handles) == 1 and len(hipk._CACHE) == 1
    A.values().mul_(2.0)                                                  # in-place change => new handle
    assert id(hipk.handle_for(A)) not in handles


def test_c_abi_error_paths(hipk):
    L = hipk.lib()
    x = torch.zeros(64, dtype=torch.float64, device=DEV)
    out = torch.zeros(1, dtype=torch.float64, device=DEV)
    sc = hipk.scratch(DEV)
    assert L.hipk_dot(63, x.data_ptr() + 8, x.data_ptr(), hipk.HIPK_F64, out.data_ptr(), sc.data_ptr(), None) == -3   # HIPK_ERR_ALIGN
    assert b"aligned" in L.hipk_last_error()
    assert L.hipk_dot(-1, x.data_ptr(), x.data_ptr(), hipk.HIPK_F64, out.data_ptr(), sc.data_ptr(), None) == -1       # HIPK_ERR_ARG
    assert L.hipk_dot(64, x.data_ptr(), x.data_ptr(), 7, out.data_ptr(), sc.data_ptr(), None) == -4                   # HIPK_ERR_UNSUPPORTED
    h = ctypes.c_void_p()
    crow = torch.tensor([0, 1, 2], device=DEV)
    col = torch.tensor([0, 1], device=DEV)
    val = torch.ones(2, dtype=torch.float64, device=DEV)
    assert L.hipk_csr_create(ctypes.byref(h), 2, 2, 2, crow.data_ptr(), col.data_ptr(), 3, val.data_ptr(), 1, None) == -1  # idx_bytes
    assert L.hipk_csr_create(ctypes.byref(h), 2, 2, 2, crow.data_ptr(), col.data_ptr(), 8, val.data_ptr(), 1, None) == 0
    prm, st = hipk.Params(), hipk.Stats()
    prm.tol, prm.maxiter = 1e-6, -1
    b = torch.ones(2, dtype=torch.float64, device=DEV)
    xs = torch.zeros(2, dtype=torch.float64, device=DEV)
    work = torch.empty(16, dtype=torch.uint8, device=DEV)
    assert L.hipk_cg_solve(h, b.data_ptr(), xs.data_ptr(), work.data_ptr(), 16, ctypes.byref(prm), ctypes.byref(st), None) == -5  # workspace
    prm.restart = 256
    wb = L.hipk_gmres_work_bytes(2, 255, 1)
    work = torch.empty(wb, dtype=torch.uint8, device=DEV)
    assert L.hipk_gmres_solve(h, b.data_ptr(), xs.data_ptr(), work.data_ptr(), wb, ctypes.byref(prm), ctypes.byref(st), None) == -4  # restart > 255
    prm.restart = 40
    assert L.hipk_gmres_solve(h, b.data_ptr(), xs.data_ptr(), work.data_ptr(), 1024, ctypes.byref(prm), ctypes.byref(st), None) == -5  # workspace
    assert L.hipk_cg_solve(h, b.data_ptr(), b.data_ptr(), work.data_ptr(), wb, ctypes.byref(prm), ctypes.byref(st), None) == -1  # aliasing
    L.hipk_csr_destroy(h)
    # shape errors through the Python surface keep the reference's exception types
    from pytorch_sparse_solver.module_a import cg
    with pytest.raises(ValueError, match="square matrix"):
        cg(torch.zeros(3, 4, dtype=torch.float64, device=DEV), torch.zeros(4, dtype=torch.float64, device=DEV))
    with pytest.raises(ValueError, match="matching shapes"):
        cg(torch.eye(4, dtype=torch.float64, device=DEV), torch.ones(4, dtype=torch.float64, device=DEV),
           x0=torch.zeros(3, dtype=torch.float64, device=DEV))


def test_restart_above_31_uses_reference_semantics(hipk):
    """restart 32 ... 255 stays on the HIP kernels (H in a workspace block), beyond that the generic path: both must work."""
    from pytorch_sparse_solver.module_a import get_last_stats, gmres
    A, g = _spd(60)
    b = torch.randn(60, dtype=torch.float64, generator=g).to(DEV)
    for restart in (40, 300):
        x, info = gmres(A, b, tol=1e-10, restart=restart)
        assert info == 0 and (torch.norm(b - A @ x) / torch.norm(b)).item() < 1e-8
    # restart = n (full GMRES) and restart > n on a small device system
    for restart in (60, 100, 255):
        for method in ("batched", "incremental"):
            x, info = gmres(A, b, tol=1e-10, restart=restart, solve_method=method)
            assert info == 0 and (torch.norm(b - A @ x) / torch.norm(b)).item() < 1e-8


@pytest.mark.gpu
def test_gmres_small_system_launch_folding_is_bit_identical(monkeypatch):
    """Systems of <= 8 reduction chunks fold the CGS2 decision and the h-vector reduction into their consumers
    (7 instead of 10 launches per Arnoldi step): same bits as the general launch sequence."""
    import torch
    from pytorch_sparse_solver.module_a import get_last_stats, gmres
    from pytorch_sparse_solver.utils.matrix_utils import create_convdiff_2d_csr, create_ldc_pressure_csr
    for A, kw in ((create_ldc_pressure_csr(100, device="cuda:0"), dict(tol=1e-10, restart=30, maxiter=30)),
                  (create_convdiff_2d_csr(90, 90, device="cuda:0"), dict(tol=1e-9, restart=12, maxiter=40, solve_method="incremental"))):
        n = A.shape[0]
        b = torch.randn(n, dtype=torch.float64, device="cuda:0", generator=torch.Generator(device="cuda:0").manual_seed(n))
        if kw["restart"] == 30:
            b -= b.mean()                                      # the Neumann pressure matrix is singular: compatible rhs
        out = {}
        for flag in ("0", "1"):
            if flag == "1":
                monkeypatch.setenv("HIPK_GMRES_NO_SMALL", "1")
            else:
                monkeypatch.delenv("HIPK_GMRES_NO_SMALL", raising=False)
            x, info = gmres(A, b, **kw)
            st = get_last_stats()
            out[flag] = (x.clone(), info, st.iterations, st.matvecs, st.residual_norm)
        assert torch.equal(out["0"][0], out["1"][0]) and out["0"][1:] == out["1"][1:]


@pytest.mark.gpu
def test_preconditioned_entry_points_validate_their_arguments():
    import ctypes
    import torch
    from pytorch_sparse_solver import _hipk
    from pytorch_sparse_solver.utils.matrix_utils import create_poisson_2d_csr
    A = create_poisson_2d_csr(12, 12, device="cuda:0")
    h = _hipk.handle_for(A)
    L = _hipk.lib()
    n = 144
    b = torch.ones(n, dtype=torch.float64, device="cuda:0")
    x = torch.zeros_like(b)
    d = torch.full((n,), 0.25, dtype=torch.float64, device="cuda:0")
    prm, st = _hipk.Params(), _hipk.Stats()
    prm.tol, prm.maxiter, prm.restart = 1e-8, -1, 10
    s = torch.cuda.current_stream().cuda_stream
    for fn, wb in ((L.hipk_pcg_solve, L.hipk_pcg_work_bytes(n, _hipk.HIPK_F64)),
                   (L.hipk_pbicgstab_solve, L.hipk_pbicgstab_work_bytes(n, _hipk.HIPK_F64)),
                   (L.hipk_pgmres_solve, L.hipk_gmres_work_bytes(n, 10, _hipk.HIPK_F64))):
        work = torch.empty(int(wb), dtype=torch.uint8, device="cuda:0")
        args = lambda **o: [o.get("h", h.ptr), o.get("d", d.data_ptr()), b.data_ptr(), o.get("x", x.data_ptr()), work.data_ptr(),
                            o.get("wb", int(wb)), ctypes.byref(prm), ctypes.byref(st), s]
        assert fn(*args()) == 0 and st.info == 0
        assert fn(*args(d=None)) == _hipk.HIPK_ERR_ARG if hasattr(_hipk, "HIPK_ERR_ARG") else fn(*args(d=None)) < 0
        assert fn(*args(wb=16)) < 0 and b"work too small" in L.hipk_last_error()
        assert fn(*args(x=b.data_ptr())) < 0                      # b and x alias
        assert fn(*args(d=d.data_ptr() + 8)) < 0                  # misaligned dinv


@pytest.mark.gpu
def test_handle_create_destroy_does_not_leak_device_memory():
    """All three handle flavours (plain, pair-coded, offset-coded) release what they allocate."""
    import gc
    import torch
    from pytorch_sparse_solver import _hipk
    from pytorch_sparse_solver.utils.matrix_utils import create_poisson_2d_csr, create_variable_diffusion_2d_csr
    mats = [create_poisson_2d_csr(700, 700, device="cuda:0"), create_variable_diffusion_2d_csr(700, 700, device="cuda:0")]
    x = torch.ones(490_000, dtype=torch.float64, device="cuda:0")

    def cycle(n):
        for k in range(n):
            A = mats[k % 2]
            h = _hipk.CsrHandle(A.crow_indices(), A.col_indices(), A.values(), A.shape)
            _hipk.spmv(h, x)
            h.close()
    cycle(4)
    gc.collect()
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    cycle(60)
    gc.collect()
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 64 << 20, (free0, free1)


@pytest.mark.gpu
def test_cg_small_system_path_without_combine_launch_is_bit_identical(monkeypatch):
    """Systems of <= 8 reduction chunks: the CG vector kernels fold the SpMV's tile sums themselves (3 instead of 4
    launches per iteration) -- same bits as the general launch sequence, on the coded, tile and row-per-wavefront paths."""
    import torch
    from pytorch_sparse_solver.module_a import cg, get_last_stats
    from pytorch_sparse_solver.utils.matrix_utils import create_poisson_2d_csr, create_variable_diffusion_2d_csr
    g = torch.Generator().manual_seed(0)
    G = torch.randn(300, 300, dtype=torch.float64, generator=g)
    dense = (G @ G.T + 300 * torch.eye(300, dtype=torch.float64)).to("cuda:0")      # dense-as-CSR: row per wavefront
    for A in (create_poisson_2d_csr(100, 100, device="cuda:0"), create_variable_diffusion_2d_csr(90, 70, device="cuda:0"),
              create_poisson_2d_csr(3, 3, device="cuda:0"), dense):
        n = A.shape[0]
        b = torch.randn(n, dtype=torch.float64, device="cuda:0", generator=torch.Generator(device="cuda:0").manual_seed(n))
        out = {}
        for flag in ("0", "1"):
            if flag == "1":
                monkeypatch.setenv("HIPK_CG_NO_SMALL", "1")
            else:
                monkeypatch.delenv("HIPK_CG_NO_SMALL", raising=False)
            x, info = cg(A, b, tol=1e-10)
            st = get_last_stats()
            out[flag] = (x.clone(), info, st.iterations, st.residual_norm, st.recurrence_rs)
        assert torch.equal(out["0"][0], out["1"][0]) and out["0"][1:] == out["1"][1:] and out["0"][1] == 0


@pytest.mark.gpu
def test_bicgstab_small_system_path_without_combine_launches_is_bit_identical(monkeypatch):
    import torch
    from pytorch_sparse_solver.module_a import JacobiPreconditioner, bicgstab, get_last_stats
    from pytorch_sparse_solver.utils.matrix_utils import create_convdiff_2d_csr, create_ldc_pressure_csr
    for A, M in ((create_convdiff_2d_csr(100, 100, device="cuda:0"), False), (create_ldc_pressure_csr(64, device="cuda:0"), False),
                 (create_convdiff_2d_csr(60, 90, device="cuda:0"), True)):
        n = A.shape[0]
        b = torch.randn(n, dtype=torch.float64, device="cuda:0", generator=torch.Generator(device="cuda:0").manual_seed(n))
        b -= b.mean()
        kw = dict(M=JacobiPreconditioner(A)) if M else {}
        out = {}
        for flag in ("0", "1"):
            if flag == "1":
                monkeypatch.setenv("HIPK_BICGSTAB_NO_SMALL", "1")
            else:
                monkeypatch.delenv("HIPK_BICGSTAB_NO_SMALL", raising=False)
            x, info = bicgstab(A, b, tol=1e-9, maxiter=400, **kw)
            st = get_last_stats()
            out[flag] = (x.clone(), info, st.iterations, st.matvecs, st.residual_norm, st.breakdown)
        assert torch.equal(out["0"][0], out["1"][0]) and out["0"][1:] == out["1"][1:]


@pytest.mark.gpu
def test_host_signal_pacing_and_stream_polling_agree(monkeypatch):
    """The loop's deciding kernel reports progress / the stop to a pinned host word (hipk_pacer); with
    HIPK_HOST_SIGNAL=0 the host follows the loop with stream-ordered reads of the device stop word instead.
    Same x bits, same counts, for stops at iteration 0, short solves, maxiter cut-offs and long solves."""
    import torch
    from pytorch_sparse_solver.module_a import JacobiPreconditioner, bicgstab, cg, get_last_stats
    from pytorch_sparse_solver.utils.matrix_utils import (create_convdiff_2d_csr, create_poisson_2d_csr,
                                                          create_variable_diffusion_2d_csr)
    dev = "cuda:0"
    cases = []
    for nx, kw in ((10, dict(tol=1e-8)), (10, dict(tol=1e-8, maxiter=3)), (10, dict(tol=1e-8, maxiter=0)),
                   (10, dict(tol=1.0)), (150, dict(tol=1e-10)), (300, dict(tol=1e-6, maxiter=70))):
        cases.append((cg, create_poisson_2d_csr(nx, nx, device=dev), kw, None))
        cases.append((bicgstab, create_convdiff_2d_csr(nx, nx, device=dev), kw, None))
    Av = create_variable_diffusion_2d_csr(60, 50, device=dev)
    cases.append((cg, Av, dict(tol=1e-9), "jacobi"))
    cases.append((bicgstab, Av, dict(tol=1e-9), "jacobi"))
    for f, A, kw, pre in cases:
        n = A.shape[0]
        b = torch.randn(n, dtype=torch.float64, device=dev, generator=torch.Generator(device=dev).manual_seed(n))
        if pre:
            kw = dict(kw, M=JacobiPreconditioner(A))
        out = {}
        for flag in ("1", "0"):
            monkeypatch.setenv("HIPK_HOST_SIGNAL", flag)
            x, info = f(A, b, **kw)
            st = get_last_stats()
            out[flag] = (x.clone(), info, st.iterations, st.matvecs, st.residual_norm, st.recurrence_rs, st.breakdown)
        assert torch.equal(out["0"][0], out["1"][0]) and out["0"][1:] == out["1"][1:], (f.__name__, n, kw, out["0"][1:], out["1"][1:])
    monkeypatch.delenv("HIPK_HOST_SIGNAL", raising=False)
    # the mid-solve fallback: the host gives up on the signal word at once (time-out 0) and carries on with
    # stream-ordered polling while the device keeps signalling -- same bits again
    from pytorch_sparse_solver.utils.matrix_utils import create_poisson_2d_csr as mk
    A = mk(300, 300, device=dev)
    b = torch.randn(90000, dtype=torch.float64, device=dev, generator=torch.Generator(device=dev).manual_seed(9))
    for f in (cg, bicgstab):
        monkeypatch.delenv("HIPK_PACE_TIMEOUT_US", raising=False)
        x0_, i0_ = f(A, b, tol=1e-9)
        s0_ = get_last_stats()
        monkeypatch.setenv("HIPK_PACE_TIMEOUT_US", "0")
        x1_, i1_ = f(A, b, tol=1e-9)
        s1_ = get_last_stats()
        assert torch.equal(x0_, x1_) and (i0_, s0_.iterations, s0_.matvecs) == (i1_, s1_.iterations, s1_.matvecs)
    monkeypatch.delenv("HIPK_PACE_TIMEOUT_US", raising=False)


@pytest.mark.gpu
def test_gmres_wide_small_system_kernels_are_bit_identical(monkeypatch):
    """Small systems with 2048-row chunks run the GMRES multi-dot / update with one 256-thread group per chunk-loop
    step (1024 threads fp64, 512 fp32); HIPK_GMRES_NO_WIDE=1 selects the 256-thread kernels: same bits, ragged tails,
    second CGS passes and fp32 storage included."""
    import torch
    from pytorch_sparse_solver.module_a import gmres, get_last_stats
    from pytorch_sparse_solver.utils.matrix_utils import create_convdiff_2d_csr, create_ldc_pressure_csr
    dev = "cuda:0"
    mats = [create_convdiff_2d_csr(100, 100, device=dev), create_ldc_pressure_csr(47, device=dev),
            create_convdiff_2d_csr(7, 5, device=dev), create_convdiff_2d_csr(128, 128, device=dev)]
    for A in mats:
        n = A.shape[0]
        for dt in (torch.float64, torch.float32):
            Ad = A if dt == torch.float64 else torch.sparse_csr_tensor(A.crow_indices(), A.col_indices(),
                                                                        A.values().float(), size=A.shape)
            b = torch.randn(n, dtype=dt, device=dev, generator=torch.Generator(device=dev).manual_seed(n))
            for method in ("batched", "incremental"):
                out = {}
                for flag in ("0", "1"):
                    if flag == "1":
                        monkeypatch.setenv("HIPK_GMRES_NO_WIDE", "1")
                    else:
                        monkeypatch.delenv("HIPK_GMRES_NO_WIDE", raising=False)
                    x, info = gmres(Ad, b, tol=1e-9 if dt == torch.float64 else 1e-4, restart=30, maxiter=6,
                                    solve_method=method)
                    st = get_last_stats()
                    out[flag] = (x.clone(), info, st.iterations, st.matvecs, st.residual_norm)
                assert torch.equal(out["0"][0], out["1"][0]) and out["0"][1:] == out["1"][1:], (n, dt, method)
    monkeypatch.delenv("HIPK_GMRES_NO_WIDE", raising=False)


def _banded_csr(n, half, dev, dtype):
    """Nonsymmetric diagonally dominant band matrix with 2*half+1 entries per interior row (rows longer than the 16 entries
    the LDS cycle kernel keeps in registers)."""
    import torch
    offs = torch.arange(-half, half + 1)
    rows = torch.arange(n).repeat_interleave(offs.numel())
    cols = rows + offs.repeat(n)
    keep = (cols >= 0) & (cols < n)
    rows, cols = rows[keep], cols[keep]
    g = torch.Generator().manual_seed(n + half)
    vals = torch.rand(rows.numel(), generator=g, dtype=torch.float64) - 0.3
    vals[rows == cols] = 2.0 * half + 1.0
    A = torch.sparse_coo_tensor(torch.stack([rows, cols]), vals, (n, n)).coalesce().to_sparse_csr()
    return torch.sparse_csr_tensor(A.crow_indices(), A.col_indices(), A.values().to(dtype), size=(n, n)).to(dev)


@pytest.mark.gpu
def test_gmres_one_launch_per_cycle_kernel_is_bit_identical(monkeypatch):
    """small systems with short rows run whole restart cycles -- the whole solve -- in ONE launch: by
    default hipk_gm_solve_lds_kernel (basis resident in LDS, eight workgroups per reduction chunk, sub-partials folded with
    the last three levels of the spec's tree; least squares, x update, residual and loop test on the device), with
    HIPK_GMRES_NO_LDS_CYCLE=1 hipk_gm_cycle_small_kernel (one workgroup per chunk, one launch per cycle);
    HIPK_GMRES_NO_CYCLE=1 selects the multi-launch small-system path; HIPK_GM_LAUNCH_CYCLES bounds the cycles of one launch
    and HIPK_GM_CYCLE_AGENT=1 selects agent-scope hand-offs.  Same bits from all of them -- ragged tails, one to
    eight chunks, second CGS passes, breakdown / early exit, Jacobi scaling, fp32 storage, rows beyond the register-held
    16 entries and restart lengths on either side of a multiple of 8 included."""
    import torch
    from pytorch_sparse_solver.module_a import JacobiPreconditioner, get_last_stats, gmres
    from pytorch_sparse_solver.utils.matrix_utils import (create_convdiff_2d_csr, create_ldc_pressure_csr,
                                                          create_variable_diffusion_2d_csr)
    dev = "cuda:0"
    mats = [create_convdiff_2d_csr(100, 100, device=dev), create_ldc_pressure_csr(100, device=dev),
            create_convdiff_2d_csr(7, 5, device=dev), create_convdiff_2d_csr(128, 128, device=dev),
            create_ldc_pressure_csr(47, device=dev), create_variable_diffusion_2d_csr(90, 70, device=dev),
            _banded_csr(5000, 12, dev, torch.float64), _banded_csr(2049, 9, dev, torch.float64),
            # 9 .. 32 chunks: the same kernel spread over the chip (agent-scope hand-offs, a lane per chunk in the folds)
            create_convdiff_2d_csr(181, 181, device=dev), create_ldc_pressure_csr(150, device=dev),
            create_convdiff_2d_csr(256, 256, device=dev), _banded_csr(40000, 10, dev, torch.float64)]
    restarts = {2: 31, 3: 7, 4: 1, 6: 17, 7: 9, 9: 20, 11: 12}   # 31: the longest cycle the LDS kernel holds; 1: a cycle of one step
    variants = ({}, {"HIPK_GMRES_NO_LDS_CYCLE": "1"}, {"HIPK_GMRES_NO_CYCLE": "1"}, {"HIPK_GM_LAUNCH_CYCLES": "1"},
                {"HIPK_GM_CYCLE_AGENT": "1", "HIPK_GM_LAUNCH_CYCLES": "2"})
    for mi, A in enumerate(mats):
        n = A.shape[0]
        for dt in (torch.float64, torch.float32):
            Ad = A if dt == torch.float64 else torch.sparse_csr_tensor(A.crow_indices(), A.col_indices(),
                                                                        A.values().float(), size=A.shape)
            b = torch.randn(n, dtype=dt, device=dev, generator=torch.Generator(device=dev).manual_seed(n))
            for method, M in (("batched", None), ("incremental", None), ("batched", "jacobi")):
                if M == "jacobi" and mi not in (0, 5, 8):
                    continue
                kw = dict(tol=1e-9 if dt == torch.float64 else 1e-4, restart=restarts.get(mi, 30), maxiter=6, solve_method=method)
                if M == "jacobi":
                    kw["M"] = JacobiPreconditioner(Ad)
                out = []
                for env in variants:
                    for key in ("HIPK_GMRES_NO_LDS_CYCLE", "HIPK_GMRES_NO_CYCLE", "HIPK_GM_LAUNCH_CYCLES", "HIPK_GM_CYCLE_AGENT"):
                        monkeypatch.delenv(key, raising=False)
                    for key, v in env.items():
                        monkeypatch.setenv(key, v)
                    x, info = gmres(Ad, b, **kw)
                    st = get_last_stats()
                    out.append((x.clone(), info, st.iterations, st.matvecs, st.residual_norm))
                for o in out[1:]:
                    assert torch.equal(out[0][0], o[0]) and out[0][1:] == o[1:], (n, dt, method, M)
    for key in ("HIPK_GMRES_NO_LDS_CYCLE", "HIPK_GMRES_NO_CYCLE", "HIPK_GM_LAUNCH_CYCLES", "HIPK_GM_CYCLE_AGENT"):
        monkeypatch.delenv(key, raising=False)
    # a tiny exactly solvable system: happy breakdown inside the cycle kernel
    A = torch.eye(5, dtype=torch.float64, device=dev).to_sparse_csr()
    x, info = gmres(A, torch.arange(1.0, 6.0, dtype=torch.float64, device=dev), tol=1e-12, restart=5)
    assert info == 0 and torch.allclose(x, torch.arange(1.0, 6.0, dtype=torch.float64, device=dev))


@pytest.mark.gpu
def test_cg_whole_loop_in_one_launch_is_bit_identical(monkeypatch):
    """Small systems with short rows run the whole CG loop in ONE launch (hipk_cg_solve_lds_kernel: x, r, p in registers, eight
    workgroups per reduction chunk, two hand-offs per iteration, p advanced on the consumer side).  HIPK_CG_NO_LDS_LOOP=1 selects
    the three-launches-per-iteration loop: same bits, same counts -- ragged tails, one to eight chunks, x0, maxiter cut-offs that
    fall inside and on the boundary of a launch's iteration budget, agent-scope hand-offs, fp32 storage."""
    import torch
    from pytorch_sparse_solver.module_a import JacobiPreconditioner, cg, get_last_stats
    from pytorch_sparse_solver.utils.matrix_utils import (create_ldc_pressure_csr, create_poisson_2d_csr,
                                                          create_variable_diffusion_2d_csr)
    dev = "cuda:0"
    mats = [create_poisson_2d_csr(100, 100, device=dev), create_poisson_2d_csr(7, 5, device=dev),
            create_poisson_2d_csr(128, 128, device=dev), create_variable_diffusion_2d_csr(90, 70, device=dev),
            create_poisson_2d_csr(45, 46, device=dev), create_ldc_pressure_csr(64, device=dev),
            # more than 8 chunks: up to 512 workgroups spread over the chip (agent-scope hand-offs, one lane per chunk in the folds)
            create_poisson_2d_csr(181, 181, device=dev), create_poisson_2d_csr(300, 200, device=dev),
            create_poisson_2d_csr(256, 256, device=dev), create_variable_diffusion_2d_csr(129, 128, device=dev)]
    # (fp64 systems of 9 .. 32 chunks take the one-workgroup-per-chunk loop of hipk_cg_mid.h by default; HIPK_CG_MID=0 keeps this
    # kernel on them)
    variants = ({}, {"HIPK_CG_NO_LDS_LOOP": "1"}, {"HIPK_CG_LAUNCH_ITS": "7"}, {"HIPK_CG_LOOP_AGENT": "1", "HIPK_CG_LAUNCH_ITS": "50"},
                {"HIPK_CG_MID": "0"}, {"HIPK_CG_MID": "0", "HIPK_CG_LAUNCH_ITS": "7"})
    keys = ("HIPK_CG_NO_LDS_LOOP", "HIPK_CG_LAUNCH_ITS", "HIPK_CG_LOOP_AGENT", "HIPK_CG_MID")
    for mi, A in enumerate(mats):
        n = A.shape[0]
        for dt in (torch.float64, torch.float32):
            Ad = A if dt == torch.float64 else torch.sparse_csr_tensor(A.crow_indices(), A.col_indices(),
                                                                        A.values().float(), size=A.shape)
            g = torch.Generator(device=dev).manual_seed(n)
            b = torch.randn(n, dtype=dt, device=dev, generator=g)
            x0 = torch.randn(n, dtype=dt, device=dev, generator=g)
            jac = JacobiPreconditioner(Ad)   # the same loop with M = diag(A)^-1 (hipk_cg_solve_lds_kernel<.., PRE>)
            for kw in (dict(tol=1e-8 if dt == torch.float64 else 1e-4), dict(tol=1e-12, maxiter=21), dict(tol=1e-12, maxiter=7),
                       dict(tol=1e-6, x0=x0), dict(tol=1e-30, maxiter=0), dict(tol=1e-8 if dt == torch.float64 else 1e-4, M=jac),
                       dict(tol=1e-12, maxiter=7, M=jac), dict(tol=1e-6, x0=x0, M=jac)):
                out = []
                for env in variants:
                    for key in keys:
                        monkeypatch.delenv(key, raising=False)
                    for key, v in env.items():
                        monkeypatch.setenv(key, v)
                    x, info = cg(Ad, b, **kw)
                    st = get_last_stats()
                    out.append((x.clone(), info, st.iterations, st.matvecs, st.residual_norm, st.recurrence_rs))
                for o in out[1:]:
                    assert torch.equal(out[0][0], o[0]) and out[0][1:] == o[1:], (n, dt, kw.keys())
    for key in keys:
        monkeypatch.delenv(key, raising=False)


@pytest.mark.gpu
def test_bicgstab_whole_loop_in_one_launch_is_bit_identical(monkeypatch):
    """Small systems with short rows run the whole BiCGStab loop in ONE launch (hipk_bi_solve_lds_kernel: three hand-offs per
    iteration, p and s advanced on the consumer side); HIPK_BICGSTAB_NO_LDS_LOOP=1 selects the five-launches-per-iteration loop:
    same bits, same counts and the same breakdown codes -- ragged tails, one to eight chunks, x0, maxiter cut-offs inside and on
    the boundary of a launch's iteration budget, the early exit, agent-scope hand-offs, fp32 storage."""
    import torch
    from pytorch_sparse_solver.module_a import JacobiPreconditioner, bicgstab, get_last_stats
    from pytorch_sparse_solver.utils.matrix_utils import (create_convdiff_2d_csr, create_ldc_pressure_csr, create_poisson_2d_csr,
                                                          create_variable_diffusion_2d_csr)
    dev = "cuda:0"
    mats = [create_convdiff_2d_csr(100, 100, device=dev), create_convdiff_2d_csr(7, 5, device=dev),
            create_poisson_2d_csr(128, 128, device=dev), create_variable_diffusion_2d_csr(90, 70, device=dev),
            create_convdiff_2d_csr(45, 46, device=dev), create_ldc_pressure_csr(64, device=dev),
            torch.eye(300, dtype=torch.float64, device=dev).to_sparse_csr(),
            # more than 8 chunks: workgroups spread over the chip (agent-scope hand-offs, one lane per chunk in the folds)
            create_convdiff_2d_csr(181, 181, device=dev), create_convdiff_2d_csr(300, 200, device=dev),
            create_convdiff_2d_csr(256, 256, device=dev)]
    # (fp64 systems of 9 .. 32 chunks with M = identity take the one-workgroup-per-chunk loop of hipk_bi_mid.h by default;
    # HIPK_BICGSTAB_MID=0 keeps this kernel on them)
    variants = ({}, {"HIPK_BICGSTAB_NO_LDS_LOOP": "1"}, {"HIPK_BICGSTAB_LAUNCH_ITS": "5"},
                {"HIPK_BICGSTAB_LOOP_AGENT": "1", "HIPK_BICGSTAB_LAUNCH_ITS": "40"}, {"HIPK_BICGSTAB_MID": "0"},
                {"HIPK_BICGSTAB_MID": "0", "HIPK_BICGSTAB_LAUNCH_ITS": "5"})
    keys = ("HIPK_BICGSTAB_NO_LDS_LOOP", "HIPK_BICGSTAB_LAUNCH_ITS", "HIPK_BICGSTAB_LOOP_AGENT", "HIPK_BICGSTAB_MID")
    for mi, A in enumerate(mats):
        n = A.shape[0]
        for dt in (torch.float64, torch.float32):
            Ad = A if dt == torch.float64 else torch.sparse_csr_tensor(A.crow_indices(), A.col_indices(),
                                                                        A.values().float(), size=A.shape)
            g = torch.Generator(device=dev).manual_seed(n)
            b = torch.randn(n, dtype=dt, device=dev, generator=g)
            x0 = torch.randn(n, dtype=dt, device=dev, generator=g)
            jac = JacobiPreconditioner(Ad)   # M = diag(A)^-1 before A (TSL:908, 922): hipk_bi_solve_lds_kernel<.., PRE>
            for kw in (dict(tol=1e-8 if dt == torch.float64 else 1e-4), dict(tol=1e-12, maxiter=15), dict(tol=1e-12, maxiter=5),
                       dict(tol=1e-6, x0=x0), dict(tol=1e-30, maxiter=0), dict(tol=1e-8 if dt == torch.float64 else 1e-4, M=jac),
                       dict(tol=1e-12, maxiter=5, M=jac), dict(tol=1e-6, x0=x0, M=jac)):
                out = []
                for env in variants:
                    for key in keys:
                        monkeypatch.delenv(key, raising=False)
                    for key, v in env.items():
                        monkeypatch.setenv(key, v)
                    x, info = bicgstab(Ad, b, **kw)
                    st = get_last_stats()
                    out.append((x.clone(), info, st.iterations, st.matvecs, st.residual_norm, st.recurrence_rs, st.breakdown))
                for o in out[1:]:   # a solve that diverges to NaN must do so in both (the sign bit of a NaN is not specified)
                    assert torch.equal(torch.isnan(out[0][0]), torch.isnan(o[0])), (n, dt, kw.keys())
                    assert torch.equal(torch.nan_to_num(out[0][0], nan=0.5), torch.nan_to_num(o[0], nan=0.5)), (n, dt, kw.keys())
                    assert all(a == b or (a != a and b != b) for a, b in zip(out[0][1:], o[1:])), (n, dt, kw.keys(), out[0][1:], o[1:])
    for key in keys:
        monkeypatch.delenv(key, raising=False)


@pytest.mark.gpu
def test_one_launch_kernels_fall_back_when_their_workgroups_are_not_resident(monkeypatch):
    """The whole-solve kernels of small systems need all their workgroups resident at once; when the placement check fails
    (a shared device) nothing has been modified and the launch sequences take over.  HIPK_TEST_LDS_NOT_RESIDENT makes the
    kernels report exactly that: same results as without them."""
    import torch
    from pytorch_sparse_solver.module_a import bicgstab, cg, get_last_stats, gmres
    from pytorch_sparse_solver.utils.matrix_utils import create_convdiff_2d_csr, create_poisson_2d_csr
    dev = "cuda:0"
    for fn, A, kw in ((cg, create_poisson_2d_csr(60, 50, device=dev), dict(tol=1e-9)),
                      (bicgstab, create_convdiff_2d_csr(60, 50, device=dev), dict(tol=1e-9)),
                      (gmres, create_convdiff_2d_csr(60, 50, device=dev), dict(tol=1e-9, restart=20, maxiter=30)),
                      (gmres, create_convdiff_2d_csr(60, 50, device=dev), dict(tol=1e-9, restart=20, maxiter=30, solve_method="incremental"))):
        n = A.shape[0]
        b = torch.randn(n, dtype=torch.float64, device=dev, generator=torch.Generator(device=dev).manual_seed(n))
        out = []
        for flag in (None, "1"):
            if flag:
                monkeypatch.setenv("HIPK_TEST_LDS_NOT_RESIDENT", flag)
            else:
                monkeypatch.delenv("HIPK_TEST_LDS_NOT_RESIDENT", raising=False)
            x, info = fn(A, b, **kw)
            st = get_last_stats()
            out.append((x.clone(), info, st.iterations, st.matvecs, st.residual_norm))
        assert out[0][1] == 0 and torch.equal(out[0][0], out[1][0]) and out[0][1:] == out[1][1:], (fn.__name__, out[0][1:], out[1][1:])
    monkeypatch.delenv("HIPK_TEST_LDS_NOT_RESIDENT", raising=False)


@pytest.mark.gpu
def test_one_launch_kernels_fall_back_in_the_middle_of_a_solve(monkeypatch):
    """a LATER launch of a one-launch loop may find its workgroups not co-resident (the first ran, e.g. the
    16384-iteration bound was reached or HIPK_*_LAUNCH_ITS is set).  The launch sequences then take over from iteration it > 0,
    where the one-launch kernels keep their reduction state in another layout (Jacobi-PCG: <r,z> only as a scalar; BiCGStab: 8 g
    sub-partials).  Launches bounded to 7 iterations / 2 cycles, the SECOND launch fails: same bits as an undisturbed solve."""
    import torch
    from pytorch_sparse_solver.module_a import JacobiPreconditioner, bicgstab, cg, get_last_stats, gmres
    from pytorch_sparse_solver.utils.matrix_utils import create_convdiff_2d_csr, create_poisson_2d_csr
    dev = "cuda:0"
    P, C = create_poisson_2d_csr(60, 50, device=dev), create_convdiff_2d_csr(60, 50, device=dev)
    # a variable diagonal so that the Jacobi preconditioner is not a multiple of the identity
    Pv = torch.sparse_csr_tensor(P.crow_indices(), P.col_indices(),
                                 P.values() * (1.0 + 0.25 * torch.sin(torch.arange(P.values().numel(), device=dev, dtype=torch.float64))),
                                 size=P.shape)
    Pd = (Pv.to_dense() + Pv.to_dense().T) / 2 + 2.0 * torch.eye(P.shape[0], device=dev, dtype=torch.float64)
    Ps = Pd.to_sparse_csr()
    cases = ((cg, P, dict(tol=1e-9)), (cg, Ps, dict(tol=1e-9, M=JacobiPreconditioner(Ps))),
             (bicgstab, C, dict(tol=1e-9)), (bicgstab, C, dict(tol=1e-9, M=JacobiPreconditioner(C))),
             (gmres, C, dict(tol=1e-9, restart=20, maxiter=30)),
             (gmres, C, dict(tol=1e-9, restart=20, maxiter=30, solve_method="incremental")))
    envs = {"HIPK_CG_LAUNCH_ITS": "7", "HIPK_BICGSTAB_LAUNCH_ITS": "7", "HIPK_GM_LAUNCH_CYCLES": "2",
            "HIPK_TEST_LDS_NOT_RESIDENT": "2"}
    for fn, A, kw in cases:
        n = A.shape[0]
        b = torch.randn(n, dtype=torch.float64, device=dev, generator=torch.Generator(device=dev).manual_seed(n))
        out = []
        for disturbed in (False, True):
            for k, v in envs.items():
                if disturbed:
                    monkeypatch.setenv(k, v)
                else:
                    monkeypatch.delenv(k, raising=False)
            x, info = fn(A, b, **kw)
            st = get_last_stats()
            out.append((x.clone(), info, st.iterations, st.matvecs, st.residual_norm))
        assert out[0][1] == 0 and out[0][2] > (2 if fn is gmres else 7) and torch.equal(out[0][0], out[1][0]) and out[0][1:] == out[1][1:], \
            (fn.__name__, list(kw), out[0][1:], out[1][1:])
    for k in envs:
        monkeypatch.delenv(k, raising=False)


@pytest.mark.gpu
def test_gmres_large_system_streaming_and_speculation_are_bit_identical(monkeypatch):
    """Large systems (more than 8 reduction chunks): the streaming-policy kernels (non-temporal loads of the basis columns
    beyond the resident ones) and the speculative second CGS pass (launched only where predicted; a miss is caught on the
    device and the cycle re-enqueued from that step) must return the bits of the plain launch sequence."""
    import torch
    from pytorch_sparse_solver.module_a import get_last_stats, gmres
    from pytorch_sparse_solver.utils.matrix_utils import create_convdiff_2d_csr, create_poisson_2d_csr
    dev = "cuda:0"
    monkeypatch.setenv("HIPK_GMRES_MID", "0")   # (these sizes take the one-launch step loop of hipk_gm_mid.h by default: its own test below)
    for A, kw in ((create_convdiff_2d_csr(300, 300, device=dev), dict(tol=1e-8, restart=30, maxiter=5)),
                  (create_poisson_2d_csr(200, 170, device=dev), dict(tol=1e-8, restart=12, maxiter=7, solve_method="incremental"))):
        n = A.shape[0]
        b = torch.randn(n, dtype=torch.float64, device=dev, generator=torch.Generator(device=dev).manual_seed(n))
        out = []
        for env in ({"HIPK_GMRES_NO_STREAM": "1", "HIPK_GM_SPEC": "0"}, {"HIPK_GM_SPEC": "0"}, {}, {"HIPK_GM_NRES": "0"},
                    {"HIPK_GM_NRES": "31"}, {"HIPK_GM_SPEC": "2"}):   # 2: nothing predicted -> every first need is a miss
            for k in ("HIPK_GMRES_NO_STREAM", "HIPK_GM_SPEC", "HIPK_GM_NRES"):
                monkeypatch.delenv(k, raising=False)
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            x, info = gmres(A, b, **kw)
            st = get_last_stats()
            out.append((x.clone(), info, st.iterations, st.matvecs, st.residual_norm))
        for o in out[1:]:
            assert torch.equal(o[0], out[0][0]) and o[1:] == out[0][1:]
    for k in ("HIPK_GMRES_NO_STREAM", "HIPK_GM_SPEC", "HIPK_GM_NRES", "HIPK_GMRES_MID"):
        monkeypatch.delenv(k, raising=False)


@pytest.mark.gpu
def test_threads_solving_on_one_handle_are_serialised():
    """ctypes releases the GIL during a solve; the binding's per-handle lock keeps solves that share the handle's
    scratch (tile sums, pinned signal words) from overlapping: four threads, same matrix, same bits as one thread."""
    import threading
    import torch
    from pytorch_sparse_solver.module_a import bicgstab, cg
    from pytorch_sparse_solver.utils.matrix_utils import create_poisson_2d_csr
    A = create_poisson_2d_csr(120, 120, device="cuda:0")
    n = A.shape[0]
    bs = [torch.randn(n, dtype=torch.float64, device="cuda:0", generator=torch.Generator(device="cuda:0").manual_seed(i))
          for i in range(4)]
    ref = [(cg(A, b, tol=1e-9)[0].clone(), bicgstab(A, b, tol=1e-9)[0].clone()) for b in bs]
    out, errs = [None] * 4, []

    def work(i):
        try:
            res = None
            for _ in range(5):
                res = (cg(A, bs[i], tol=1e-9)[0].clone(), bicgstab(A, bs[i], tol=1e-9)[0].clone())
            out[i] = res
        except Exception as e:  # pragma: no cover
            errs.append(e)
    th = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs
    for i in range(4):
        assert torch.equal(out[i][0], ref[i][0]) and torch.equal(out[i][1], ref[i][1])


@pytest.mark.gpu
def test_pytree_right_hand_sides_stay_on_the_hip_path(hipk):
    """VERDICT r2 "missing" 7: a device matrix with a PyTree b / x0 (TSL:186-205 applies the matrix to the concatenated leaves) is the
    flat solve: same bits as solving with the raveled vector, HIP stats (not the generic path's), the reference's structure errors."""
    from pytorch_sparse_solver.module_a import JacobiPreconditioner, bicgstab, cg, get_last_stats, gmres
    from pytorch_sparse_solver.utils.matrix_utils import create_poisson_2d_csr
    A = create_poisson_2d_csr(40, 30, device=DEV)
    n = A.shape[0]
    g = torch.Generator(device=DEV).manual_seed(4)
    bf = torch.randn(n, dtype=torch.float64, device=DEV, generator=g)
    x0f = torch.randn(n, dtype=torch.float64, device=DEV, generator=g)
    tree = lambda v: {"u": v[:500].reshape(20, 25), "v": (v[500:900], [v[900:]])}   # noqa: E731
    for fn, kw in ((cg, dict(tol=1e-7)), (bicgstab, dict(tol=1e-7)), (gmres, dict(tol=1e-7, restart=25)),
                   (cg, dict(tol=1e-7, M=JacobiPreconditioner(A)))):
        x_flat, info_flat = fn(A, bf, x0=x0f, **kw)
        st_flat = get_last_stats()
        x_tree, info_tree = fn(A, tree(bf), x0=tree(x0f), **kw)
        st = get_last_stats()
        assert type(st).__name__ == "SolveStats" and (st.iterations, st.matvecs) == (st_flat.iterations, st_flat.matvecs)
        assert info_tree == info_flat and (torch.linalg.norm(bf - A @ x_flat) <= 1e-6 * torch.linalg.norm(bf))
        assert x_tree["u"].shape == (20, 25) and torch.equal(torch.cat([x_tree["u"].reshape(-1), x_tree["v"][0], x_tree["v"][1][0]]), x_flat)
    x2, info2 = cg(A, bf.reshape(n, 1), tol=1e-7)                       # a tensor that is not 1-D is a one-leaf tree
    assert x2.shape == (n, 1) and info2 == 0
    with pytest.raises(ValueError, match="matching tree structure"):
        cg(A, tree(bf), x0={"u": x0f[:500].reshape(20, 25)})
    with pytest.raises(ValueError, match="matching shapes"):
        cg(A, tree(bf), x0={"u": x0f[:500].reshape(25, 20), "v": (x0f[500:900], [x0f[900:]])})


@pytest.mark.gpu
def test_cg_two_launch_iteration_is_bit_identical(hipk, oracle, monkeypatch):
    """Launch-bound mid-size systems (33 .. 256 reduction chunks) run CG with TWO launches per iteration: the direction step is
    formed on the fly inside the SpMV (hipk_cg2_spmv_kernel / hipk_cg2_update_kernel).  Same bits as the three-launch sequence --
    x, iteration count, info, true and recurrence residuals -- for fp64 and fp32 storage, warm starts, maxiter cut-offs (even and
    odd, 0 and 1), a stop at iteration 0; and, on one system, as the CPU oracle."""
    from pytorch_sparse_solver.module_a import get_last_stats
    from pytorch_sparse_solver.utils.matrix_utils import create_poisson_2d_csr, create_variable_diffusion_2d_csr
    cases = [(create_poisson_2d_csr(300, 300, device=DEV), dict(tol=1e-8)),                       # 44 chunks
             (create_poisson_2d_csr(500, 500, device=DEV), dict(tol=1e-6)),                       # 123 chunks
             (create_poisson_2d_csr(550, 557, device=DEV), dict(tol=1e-5)),                       # 150 chunks (the bound), ragged last chunk
             (create_variable_diffusion_2d_csr(400, 300, device=DEV), dict(tol=1e-7)),            # values differ per entry
             (create_poisson_2d_csr(300, 300, device=DEV), dict(tol=1e-12, maxiter=37)),
             (create_poisson_2d_csr(300, 300, device=DEV), dict(tol=1e-12, maxiter=38)),
             (create_poisson_2d_csr(300, 300, device=DEV), dict(tol=1e-12, maxiter=1)),
             (create_poisson_2d_csr(300, 300, device=DEV), dict(tol=1e-12, maxiter=0)),
             (create_poisson_2d_csr(300, 300, device=DEV), dict(tol=0.5))]                         # b = A x0 below: stops at iteration 0
    for idx, (A, kw) in enumerate(cases):
        for dt in (torch.float64, torch.float32):
            Ad = A if dt == torch.float64 else torch.sparse_csr_tensor(A.crow_indices(), A.col_indices(), A.values().float(), size=A.shape)
            h = hipk.handle_for(Ad)
            n = A.shape[0]
            g = torch.Generator(device=DEV).manual_seed(idx)
            b = torch.randn(n, dtype=dt, device=DEV, generator=g)
            x0 = torch.randn(n, dtype=dt, device=DEV, generator=g) if idx % 2 else None
            if idx == 8:
                x0 = torch.randn(n, dtype=dt, device=DEV, generator=g)
                b = hipk.spmv(h, x0)
            out = []
            for two in ("1", "0"):
                monkeypatch.setenv("HIPK_CG_TWO_LAUNCH", two)
                x = torch.zeros_like(b) if x0 is None else x0.clone()
                st = hipk.solve("cg", h, b, x, atol=0.0, **{"maxiter": None, **kw})
                out.append((x.clone(), st.iterations, st.matvecs, st.info, st.residual_norm, st.recurrence_rs))
            assert torch.equal(out[0][0], out[1][0]) and out[0][1:] == out[1][1:], (idx, dt, out[0][1:], out[1][1:])
            if idx == 8:
                assert out[0][1] == 0
    monkeypatch.delenv("HIPK_CG_TWO_LAUNCH", raising=False)
    A, kw = cases[0]
    b = torch.ones(A.shape[0], dtype=torch.float64, device=DEV)
    x = torch.zeros_like(b)
    st = hipk.solve("cg", hipk.handle_for(A), b, x, atol=0.0, maxiter=None, **kw)
    Ac = A.cpu()
    ref = oracle.cg(Ac.crow_indices().numpy().astype(np.int32), Ac.col_indices().numpy().astype(np.int32), Ac.values().numpy(),
                    np.ones(A.shape[0]), **kw)
    assert (st.iterations, st.info) == (ref.iterations, ref.info) and np.array_equal(x.cpu().numpy(), ref.x)


def _scipy_to_csr_dev(M):
    M = M.tocsr()
    M.sort_indices()
    return torch.sparse_csr_tensor(torch.from_numpy(M.indptr.astype(np.int64)), torch.from_numpy(M.indices.astype(np.int64)),
                                   torch.from_numpy(M.data.astype(np.float64)), size=M.shape).to(DEV)


@pytest.mark.gpu
def test_cg_mid_one_launch_is_bit_identical(hipk, oracle, monkeypatch):
    """Launch-bound mid-size fp64 systems (33 .. 512 reduction chunks, rows of <= 12 entries that stay within a window around their
    chunk) run the WHOLE CG loop in one launch, one workgroup per chunk (csrc/hipk_cg_mid.h): x and r in registers, p as an LDS window
    advanced on the consumer side, chunk partials and r exchanged as flagged words.  Same bits as the launch sequences
    (HIPK_CG_MID=0) -- x, iteration count, info, true and recurrence residuals -- for 2-D and 3-D stencils, per-entry values, a banded
    random SPD matrix with 9 .. 11 entries per row, ragged last chunks, one and two workgroups per CU, warm starts, maxiter cut-offs,
    a stop at iteration 0, re-launches every 7 iterations, a launch whose workgroups report "not co-resident" (first and second);
    and, on one system, as the CPU oracle."""
    import scipy.sparse as sp
    from pytorch_sparse_solver.utils.matrix_utils import create_poisson_2d_csr, create_variable_diffusion_2d_csr

    def poisson3d(m):
        T = sp.diags([-1.0, 2.0, -1.0], [-1, 0, 1], shape=(m, m))
        I = sp.identity(m)
        return _scipy_to_csr_dev(sp.kron(sp.kron(T, I), I) + sp.kron(sp.kron(I, T), I) + sp.kron(sp.kron(I, I), T))

    def banded_spd(n, offs, seed):
        rng = np.random.default_rng(seed)
        M = sp.diags([rng.uniform(-1.0, -0.1, n - o) for o in offs], offs, shape=(n, n))
        M = M + M.T
        return _scipy_to_csr_dev(M + sp.diags(np.asarray(abs(M).sum(axis=1)).ravel() + 0.5))

    cases = [(create_poisson_2d_csr(300, 300, device=DEV), dict(tol=1e-8), {}),                       # 44 chunks
             (create_poisson_2d_csr(500, 500, device=DEV), dict(tol=1e-6), {}),                       # 123 chunks
             (create_poisson_2d_csr(550, 557, device=DEV), dict(tol=1e-5), {}),                       # ragged last chunk
             (create_variable_diffusion_2d_csr(400, 300, device=DEV), dict(tol=1e-7), {}),            # values differ per entry
             (create_poisson_2d_csr(720, 720, device=DEV), dict(tol=1e-5), {}),                       # 254 chunks: one workgroup on (almost) every CU
             (create_poisson_2d_csr(1000, 1000, device=DEV), dict(tol=1e-4), {}),                     # 489 chunks: two per CU
             (poisson3d(48), dict(tol=1e-8), {}),                                                     # 7 entries per row, reach 2304
             (poisson3d(64), dict(tol=1e-8), {}),                                                     # 128 chunks, reach 4096: three bands of tiles
             (banded_spd(100003, (1, 2, 3, 700, 1500), 5), dict(tol=1e-10), {}),                      # 9 .. 11 entries per row
             (create_poisson_2d_csr(300, 300, device=DEV), dict(tol=1e-12, maxiter=37), {}),
             (create_poisson_2d_csr(300, 300, device=DEV), dict(tol=1e-12, maxiter=38), {}),
             (create_poisson_2d_csr(300, 300, device=DEV), dict(tol=1e-12, maxiter=1), {}),
             (create_poisson_2d_csr(300, 300, device=DEV), dict(tol=1e-12, maxiter=0), {}),
             (create_poisson_2d_csr(300, 300, device=DEV), dict(tol=0.5), {}),                        # b = A x0 below: stops at iteration 0
             (create_poisson_2d_csr(300, 300, device=DEV), dict(tol=1e-8), {"HIPK_CG_LAUNCH_ITS": "7"}),
             (create_poisson_2d_csr(300, 300, device=DEV), dict(tol=1e-8), {"HIPK_TEST_LDS_NOT_RESIDENT": "1"}),
             (create_poisson_2d_csr(300, 300, device=DEV), dict(tol=1e-8), {"HIPK_CG_LAUNCH_ITS": "7", "HIPK_TEST_LDS_NOT_RESIDENT": "2"})]
    for idx, (A, kw, env) in enumerate(cases):
      for dt in (torch.float64, torch.float32):   # fp32 storage: the kernel's T = float (tolerances it can reach, bounded iterations)
        Ad = A if dt == torch.float64 else torch.sparse_csr_tensor(A.crow_indices(), A.col_indices(), A.values().float(), size=A.shape)
        kwd = kw if dt == torch.float64 else {**kw, "tol": max(kw["tol"], 1e-4), "maxiter": min(kw.get("maxiter", 300) or 300, 300)}
        h = hipk.handle_for(Ad)
        n = A.shape[0]
        g = torch.Generator(device=DEV).manual_seed(idx)
        b = torch.randn(n, dtype=dt, device=DEV, generator=g)
        x0 = torch.randn(n, dtype=dt, device=DEV, generator=g) if idx % 2 else None
        if idx == 13:
            x0 = torch.randn(n, dtype=dt, device=DEV, generator=g)
            b = hipk.spmv(h, x0)
        out = []
        for mid in ("1", "0"):
            monkeypatch.setenv("HIPK_CG_MID", mid)
            for k, v in env.items():
                if mid == "1":
                    monkeypatch.setenv(k, v)
                else:
                    monkeypatch.delenv(k, raising=False)
            x = torch.zeros_like(b) if x0 is None else x0.clone()
            st = hipk.solve("cg", h, b, x, atol=0.0, **{"maxiter": None, **kwd})
            out.append((x.clone(), st.iterations, st.matvecs, st.info, st.residual_norm, st.recurrence_rs))
        assert torch.equal(out[0][0], out[1][0]) and out[0][1:] == out[1][1:], (idx, dt, out[0][1:], out[1][1:])
        if idx == 13:
            assert out[0][1] == 0
        if idx < 9:
            assert out[0][1] > (20 if dt == torch.float64 else 8), (idx, dt, out[0][1])
    # the same loop with M = diag(A)^-1 (hipk_cg_mid_kernel<W, 1, PRE = true>, up to 256 chunks): z = dinv .* r formed where it is
    # used, <r,z> beside <r,r>; against hipk_pcg_solve's launch sequence
    for idx, (A, kw, env) in enumerate(cases):
      if idx == 5:
          continue   # 489 chunks: not taken by the preconditioned loop either way
      for dt in (torch.float64, torch.float32):
        Ad = A if dt == torch.float64 else torch.sparse_csr_tensor(A.crow_indices(), A.col_indices(), A.values().float(), size=A.shape)
        kwd = kw if dt == torch.float64 else {**kw, "tol": max(kw["tol"], 1e-4), "maxiter": min(kw.get("maxiter", 300) or 300, 300)}
        h = hipk.handle_for(Ad)
        n = A.shape[0]
        Ac = A.cpu()
        dinv = (1.0 / torch.from_numpy(sp.csr_matrix((Ac.values().numpy(), Ac.col_indices().numpy(), Ac.crow_indices().numpy()),
                                                     shape=A.shape).diagonal())).to(DEV).to(dt)
        g = torch.Generator(device=DEV).manual_seed(100 + idx)
        b = torch.randn(n, dtype=dt, device=DEV, generator=g)
        x0 = torch.randn(n, dtype=dt, device=DEV, generator=g) if idx % 2 else None
        if idx == 13:
            x0 = torch.randn(n, dtype=dt, device=DEV, generator=g)
            b = hipk.spmv(h, x0)
        out = []
        for mid in ("1", "0"):
            monkeypatch.setenv("HIPK_CG_MID", mid)
            for k, v in env.items():
                if mid == "1":
                    monkeypatch.setenv(k, v)
                else:
                    monkeypatch.delenv(k, raising=False)
            x = torch.zeros_like(b) if x0 is None else x0.clone()
            st = hipk.solve_pcg(h, dinv, b, x, atol=0.0, **{"maxiter": None, **kwd})
            out.append((x.clone(), st.iterations, st.matvecs, st.info, st.residual_norm, st.recurrence_rs))
        assert torch.equal(out[0][0], out[1][0]) and out[0][1:] == out[1][1:], ("jacobi", idx, dt, out[0][1:], out[1][1:])
        if idx == 13:
            assert out[0][1] == 0
        if idx < 9:
            assert out[0][1] > (15 if dt == torch.float64 else 5), ("jacobi", idx, dt, out[0][1])
    monkeypatch.delenv("HIPK_CG_MID", raising=False)
    A, kw, _ = cases[0]
    b = torch.ones(A.shape[0], dtype=torch.float64, device=DEV)
    x = torch.zeros_like(b)
    st = hipk.solve("cg", hipk.handle_for(A), b, x, atol=0.0, maxiter=None, **kw)
    Ac = A.cpu()
    ref = oracle.cg(Ac.crow_indices().numpy().astype(np.int32), Ac.col_indices().numpy().astype(np.int32), Ac.values().numpy(),
                    np.ones(A.shape[0]), **kw)
    assert (st.iterations, st.info) == (ref.iterations, ref.info) and np.array_equal(x.cpu().numpy(), ref.x)


@pytest.mark.gpu
def test_bicgstab_mid_one_launch_is_bit_identical(hipk, oracle, monkeypatch):
    """Launch-bound mid-size fp64 systems (9 .. 256 reduction chunks, M = identity, rows of <= 12 entries within a window around
    their chunk) run the WHOLE BiCGStab loop in one launch, one workgroup per chunk (csrc/hipk_bi_mid.h): three hand-offs per
    iteration instead of five launches, p and s advanced over an LDS window on the consumer side.  Same bits as the launch
    sequence (HIPK_BICGSTAB_MID=0 + HIPK_BICGSTAB_NO_LDS_LOOP=1) -- x, iteration count, matvecs, info, residuals, breakdown codes --
    for nonsymmetric and symmetric stencils, per-entry values, a 3-D stencil, a banded random matrix with 9 .. 11 entries per row,
    ragged last chunks, warm starts, maxiter cut-offs, the early exit, re-launches every 5 iterations, not-co-resident launches
    (first and second), a breakdown; and, on one system, as the CPU oracle."""
    import scipy.sparse as sp
    from pytorch_sparse_solver.utils.matrix_utils import (create_convdiff_2d_csr, create_poisson_2d_csr,
                                                          create_variable_diffusion_2d_csr)

    def convdiff3d(m):
        T = sp.diags([-1.3, 2.0, -0.7], [-1, 0, 1], shape=(m, m))
        I = sp.identity(m)
        return _scipy_to_csr_dev(sp.kron(sp.kron(T, I), I) + sp.kron(sp.kron(I, T), I) + sp.kron(sp.kron(I, I), T))

    def banded(n, offs, seed):
        rng = np.random.default_rng(seed)
        M = sp.diags([rng.uniform(-1.0, -0.1, n - abs(o)) for o in offs], offs, shape=(n, n))
        return _scipy_to_csr_dev(M + sp.diags(np.asarray(abs(M).sum(axis=1)).ravel() + 0.5))

    cases = [(create_convdiff_2d_csr(300, 300, device=DEV), dict(tol=1e-8), {}),                      # 44 chunks
             (create_convdiff_2d_csr(500, 500, device=DEV), dict(tol=1e-6), {}),                      # 123 chunks
             (create_convdiff_2d_csr(550, 557, device=DEV), dict(tol=1e-5), {}),                      # ragged last chunk
             (create_variable_diffusion_2d_csr(400, 300, device=DEV), dict(tol=1e-7), {}),
             (create_poisson_2d_csr(720, 720, device=DEV), dict(tol=1e-5), {}),                       # 254 chunks
             (create_convdiff_2d_csr(150, 150, device=DEV), dict(tol=1e-9), {}),                      # 11 chunks
             (convdiff3d(32), dict(tol=1e-8), {}),                                                    # 16 chunks, 7 entries per row: planes of 1024 rows
             (banded(100003, (-1400, -700, -3, -2, -1, 1, 2, 3, 650, 1500), 5), dict(tol=1e-10), {}),
             (create_convdiff_2d_csr(300, 300, device=DEV), dict(tol=1e-12, maxiter=15), {}),
             (create_convdiff_2d_csr(300, 300, device=DEV), dict(tol=1e-12, maxiter=1), {}),
             (create_convdiff_2d_csr(300, 300, device=DEV), dict(tol=1e-12, maxiter=0), {}),
             (create_convdiff_2d_csr(300, 300, device=DEV), dict(tol=0.5), {}),                       # b = A x0 below: stops at iteration 0
             (create_convdiff_2d_csr(300, 300, device=DEV), dict(tol=1e-8), {"HIPK_BICGSTAB_LAUNCH_ITS": "5"}),
             (create_convdiff_2d_csr(300, 300, device=DEV), dict(tol=1e-8), {"HIPK_TEST_LDS_NOT_RESIDENT": "1"}),
             (create_convdiff_2d_csr(300, 300, device=DEV), dict(tol=1e-8), {"HIPK_BICGSTAB_LAUNCH_ITS": "5", "HIPK_TEST_LDS_NOT_RESIDENT": "2"}),
             (create_convdiff_2d_csr(300, 300, device=DEV), dict(tol=1e-30, maxiter=400), {})]       # runs into the iteration bound (or a breakdown)
    for idx, (A, kw, env) in enumerate(cases):
      for dt in (torch.float64, torch.float32):   # fp32 storage: the kernel's T = float (tolerances it can reach, bounded iterations)
        Ad = A if dt == torch.float64 else torch.sparse_csr_tensor(A.crow_indices(), A.col_indices(), A.values().float(), size=A.shape)
        kw = kw if dt == torch.float64 else {**kw, "tol": max(kw["tol"], 1e-4), "maxiter": min(kw.get("maxiter", 200) or 200, 200)}
        h = hipk.handle_for(Ad)
        n = A.shape[0]
        g = torch.Generator(device=DEV).manual_seed(idx)
        b = torch.randn(n, dtype=dt, device=DEV, generator=g)
        x0 = torch.randn(n, dtype=dt, device=DEV, generator=g) if idx % 2 else None
        if idx == 11:
            x0 = torch.randn(n, dtype=dt, device=DEV, generator=g)
            b = hipk.spmv(h, x0)
        out = []
        for mid in ("1", "0"):
            monkeypatch.setenv("HIPK_BICGSTAB_MID", mid)
            if mid == "0":
                monkeypatch.setenv("HIPK_BICGSTAB_NO_LDS_LOOP", "1")
            else:
                monkeypatch.delenv("HIPK_BICGSTAB_NO_LDS_LOOP", raising=False)
            for k, v in env.items():
                if mid == "1":
                    monkeypatch.setenv(k, v)
                else:
                    monkeypatch.delenv(k, raising=False)
            x = torch.zeros_like(b) if x0 is None else x0.clone()
            print("bicgstab mid case", idx, "mid" if mid == "1" else "launch sequence", flush=True)   # (-s: which case a hang is in)
            st = hipk.solve("bicgstab", h, b, x, atol=0.0, **{"maxiter": None, **kw})
            out.append((x.clone(), st.iterations, st.matvecs, st.info, st.residual_norm, st.recurrence_rs, st.breakdown))
        assert torch.equal(torch.nan_to_num(out[0][0], nan=0.5), torch.nan_to_num(out[1][0], nan=0.5)), (idx, dt, out[0][1:], out[1][1:])
        assert all(a == b_ or (a != a and b_ != b_) for a, b_ in zip(out[0][1:], out[1][1:])), (idx, dt, out[0][1:], out[1][1:])
        if idx == 11:
            assert out[0][1] == 0
        if idx < 8 and dt == torch.float64:
            assert out[0][1] > 10, (idx, out[0][1])
    # the same with M = diag(A)^-1 applied before A (hipk_bi_mid_kernel<W, PRE>: a fourth LDS window holds dinv; phat and shat formed at
    # the gathered columns); against hipk_pbicgstab_solve's launch sequence
    for idx, (A, kw, env) in enumerate(cases):
        if idx == 7:
            continue   # a window beyond the LDS with a fourth one: not taken either way
        h = hipk.handle_for(A)
        n = A.shape[0]
        Ac = A.cpu()
        dinv = (1.0 / torch.from_numpy(sp.csr_matrix((Ac.values().numpy(), Ac.col_indices().numpy(), Ac.crow_indices().numpy()),
                                                     shape=A.shape).diagonal())).to(DEV)
        g = torch.Generator(device=DEV).manual_seed(300 + idx)
        b = torch.randn(n, dtype=torch.float64, device=DEV, generator=g)
        x0 = torch.randn(n, dtype=torch.float64, device=DEV, generator=g) if idx % 2 else None
        out = []
        for mid in ("1", "0"):
            monkeypatch.setenv("HIPK_BICGSTAB_MID", mid)
            if mid == "0":
                monkeypatch.setenv("HIPK_BICGSTAB_NO_LDS_LOOP", "1")
            else:
                monkeypatch.delenv("HIPK_BICGSTAB_NO_LDS_LOOP", raising=False)
            for k, v in env.items():
                if mid == "1":
                    monkeypatch.setenv(k, v)
                else:
                    monkeypatch.delenv(k, raising=False)
            x = torch.zeros_like(b) if x0 is None else x0.clone()
            st = hipk.solve_pcg(h, dinv, b, x, atol=0.0, method="bicgstab", **{"maxiter": None, **kw})
            out.append((x.clone(), st.iterations, st.matvecs, st.info, st.residual_norm, st.recurrence_rs, st.breakdown))
        assert torch.equal(torch.nan_to_num(out[0][0], nan=0.5), torch.nan_to_num(out[1][0], nan=0.5)), ("jacobi", idx, out[0][1:], out[1][1:])
        assert all(a == b_ or (a != a and b_ != b_) for a, b_ in zip(out[0][1:], out[1][1:])), ("jacobi", idx, out[0][1:], out[1][1:])
    monkeypatch.delenv("HIPK_BICGSTAB_MID", raising=False)
    monkeypatch.delenv("HIPK_BICGSTAB_NO_LDS_LOOP", raising=False)
    A, kw, _ = cases[3]   # (the convection-diffusion systems above diverge for b = ones: a bounded run on a diffusion system)
    b = torch.ones(A.shape[0], dtype=torch.float64, device=DEV)
    x = torch.zeros_like(b)
    st = hipk.solve("bicgstab", hipk.handle_for(A), b, x, atol=0.0, maxiter=1500, **kw)
    Ac = A.cpu()
    ref = oracle.bicgstab(Ac.crow_indices().numpy().astype(np.int32), Ac.col_indices().numpy().astype(np.int32), Ac.values().numpy(),
                          np.ones(A.shape[0]), maxiter=1500, **kw)
    assert (st.iterations, st.info) == (ref.iterations, ref.info) and np.array_equal(x.cpu().numpy(), ref.x)
    assert st.iterations > 20


@pytest.mark.gpu
def test_gmres_mid_one_launch_cycle_is_bit_identical(hipk, oracle, monkeypatch):
    """Launch-bound mid-size fp64 systems (33 .. 256 reduction chunks, restart <= 31, no preconditioner, rows of <= 12 entries within
    a window around their chunk) run the Arnoldi steps of a restart cycle in ONE launch, one workgroup per chunk
    (csrc/hipk_gm_mid.h): three hand-offs per step (five with a second CGS pass) instead of five to nine launches, the basis in
    memory, v_k as an LDS window.  Same bits as the launch sequence (HIPK_GMRES_MID=0) -- x, cycles, matvecs, info, residuals,
    breakdown flag -- for both solve methods, restarts 1 .. 31, nonsymmetric and symmetric stencils, per-entry values, ragged last
    chunks, warm starts, cycle cut-offs, an early exit inside a cycle, a launch whose workgroups report "not co-resident" (first
    and second cycle); and, on one system, as the CPU oracle."""
    from pytorch_sparse_solver.utils.matrix_utils import (create_convdiff_2d_csr, create_poisson_2d_csr,
                                                          create_variable_diffusion_2d_csr)
    def grid3d(m_):   # nonsymmetric 7-point stencil on an m^3 grid
        import scipy.sparse as sp
        T = sp.diags([-1.3, 2.0, -0.7], [-1, 0, 1], shape=(m_, m_))
        I = sp.identity(m_)
        return _scipy_to_csr_dev(sp.kron(sp.kron(T, I), I) + sp.kron(sp.kron(I, T), I) + sp.kron(sp.kron(I, I), T))

    cases = [(create_convdiff_2d_csr(300, 300, device=DEV), dict(tol=1e-8, restart=30, maxiter=4), {}),          # 44 chunks
             (create_convdiff_2d_csr(300, 300, device=DEV), dict(tol=1e-8, restart=30, maxiter=4, solve_method="incremental"), {}),
             (create_poisson_2d_csr(500, 500, device=DEV), dict(tol=1e-6, restart=20, maxiter=3), {}),            # 123 chunks
             (create_poisson_2d_csr(550, 557, device=DEV), dict(tol=1e-5, restart=31, maxiter=2, solve_method="incremental"), {}),
             (create_variable_diffusion_2d_csr(400, 300, device=DEV), dict(tol=1e-7, restart=12, maxiter=6), {}),
             (create_poisson_2d_csr(720, 720, device=DEV), dict(tol=1e-5, restart=10, maxiter=3), {}),            # 254 chunks
             (create_convdiff_2d_csr(300, 300, device=DEV), dict(tol=1e-8, restart=1, maxiter=5), {}),
             (create_convdiff_2d_csr(300, 300, device=DEV), dict(tol=1e-8, restart=2, maxiter=5, solve_method="incremental"), {}),
             (create_poisson_2d_csr(300, 300, device=DEV), dict(tol=1e-3, restart=30, maxiter=50, solve_method="incremental"), {}),   # converges inside a cycle
             (create_poisson_2d_csr(300, 300, device=DEV), dict(tol=1e-3, restart=30, maxiter=50), {}),
             (create_convdiff_2d_csr(300, 300, device=DEV), dict(tol=1e-8, restart=30, maxiter=0), {}),
             (create_convdiff_2d_csr(300, 300, device=DEV), dict(tol=1e-8, restart=15, maxiter=4), {"HIPK_TEST_LDS_NOT_RESIDENT": "1"}),
             (create_convdiff_2d_csr(300, 300, device=DEV), dict(tol=1e-8, restart=15, maxiter=4), {"HIPK_TEST_LDS_NOT_RESIDENT": "2"}),
             (grid3d(64), dict(tol=1e-8, restart=25, maxiter=3), {}),                                             # 128 chunks, three bands of tiles
             (grid3d(45), dict(tol=1e-8, restart=30, maxiter=2, solve_method="incremental"), {})]               # ragged planes
    for idx, (A, kw, env) in enumerate(cases):
      for dt in (torch.float64, torch.float32):   # fp32 storage: the kernel's T = float
        Ad = A if dt == torch.float64 else torch.sparse_csr_tensor(A.crow_indices(), A.col_indices(), A.values().float(), size=A.shape)
        kwd = kw if dt == torch.float64 else {**kw, "tol": max(kw["tol"], 1e-4)}
        h = hipk.handle_for(Ad)
        n = A.shape[0]
        g = torch.Generator(device=DEV).manual_seed(idx)
        b = torch.randn(n, dtype=dt, device=DEV, generator=g)
        x0 = torch.randn(n, dtype=dt, device=DEV, generator=g) if idx % 2 else None
        out = []
        for mid in ("1", "0"):
            monkeypatch.setenv("HIPK_GMRES_MID", mid)
            for k, v in env.items():
                if mid == "1":
                    monkeypatch.setenv(k, v)
                else:
                    monkeypatch.delenv(k, raising=False)
            x = torch.zeros_like(b) if x0 is None else x0.clone()
            print("gmres mid case", idx, dt, "mid" if mid == "1" else "launch sequence", flush=True)
            st = hipk.solve("gmres", h, b, x, atol=0.0, **kwd)
            out.append((x.clone(), st.iterations, st.matvecs, st.info, st.residual_norm, st.recurrence_rs, st.breakdown))
        assert torch.equal(out[0][0], out[1][0]) and out[0][1:] == out[1][1:], (idx, dt, out[0][1:], out[1][1:])
    # the same with M = diag(d) applied after every A (hipk_gm_mid_kernel<W, PRE>: the row scaling of the SpMV epilogue); a diagonal
    # that is NOT the matrix's own, so that the scaling matters
    import scipy.sparse as sp
    for idx, (A, kw, env) in enumerate(cases):
        if idx in (6, 7, 10):
            continue
        h = hipk.handle_for(A)
        n = A.shape[0]
        g = torch.Generator(device=DEV).manual_seed(200 + idx)
        dinv = 0.2 + torch.rand(n, dtype=torch.float64, device=DEV, generator=g)
        b = torch.randn(n, dtype=torch.float64, device=DEV, generator=g)
        out = []
        for mid in ("1", "0"):
            monkeypatch.setenv("HIPK_GMRES_MID", mid)
            for k, v in env.items():
                if mid == "1":
                    monkeypatch.setenv(k, v)
                else:
                    monkeypatch.delenv(k, raising=False)
            x = torch.zeros_like(b)
            st = hipk.solve_pgmres(h, dinv, b, x, atol=0.0, **kw)
            out.append((x.clone(), st.iterations, st.matvecs, st.info, st.residual_norm, st.recurrence_rs, st.breakdown))
        assert torch.equal(out[0][0], out[1][0]) and out[0][1:] == out[1][1:], ("jacobi", idx, out[0][1:], out[1][1:])
    monkeypatch.delenv("HIPK_GMRES_MID", raising=False)
    A, kw, _ = cases[2]
    b = torch.ones(A.shape[0], dtype=torch.float64, device=DEV)
    x = torch.zeros_like(b)
    st = hipk.solve("gmres", hipk.handle_for(A), b, x, atol=0.0, **kw)
    Ac = A.cpu()
    ref = oracle.gmres(Ac.crow_indices().numpy().astype(np.int32), Ac.col_indices().numpy().astype(np.int32), Ac.values().numpy(),
                       np.ones(A.shape[0]), **kw)
    assert (st.iterations, st.info) == (ref.iterations, ref.info) and np.array_equal(x.cpu().numpy(), ref.x)
