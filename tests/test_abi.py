"""The C-ABI library loads (no GPU needed: no compute calls) and exports every symbol include/hipk.h declares."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def _declared():
    text = open(os.path.join(ROOT, "include", "hipk.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hipk_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    import torch  # noqa: F401  (HIP runtime first)
    from pytorch_sparse_solver import _hipk
    if not os.path.exists(_hipk.LIB_PATH):
        _hipk.build()
    L = _hipk.lib()
    declared = _declared()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/hipk.h but not exported by libhipk.so"
    assert sorted(_hipk.SYMBOLS) == declared, "python binding list out of sync with include/hipk.h"
    assert L.hipk_version() == 300
    assert len(L.hipk_build_id()) == 16
    assert L.hipk_scratch_bytes() == 4 * 2048 * 8
    # geometry helpers are pure host code
    assert (L.hipk_chunk_size(4_000_000), L.hipk_chunk_count(4_000_000)) == (2048, 1954)
    assert (L.hipk_chunk_size(64_000_000), L.hipk_chunk_count(64_000_000)) == (32768, 1954)
    assert L.hipk_cg_work_bytes(1000, 1) >= 3 * 8000


def test_python_geometry_matches_library_and_oracle(oracle):
    import torch  # noqa: F401
    from pytorch_sparse_solver import _hipk
    from pytorch_sparse_solver.distributed import chunk_geometry
    L = _hipk.lib()
    for n in (1, 2047, 2048, 2049, 4_000_000, 2048 * 2048, 2048 * 2048 + 1, 64_000_000, 2_000_000_000):
        assert chunk_geometry(n) == (L.hipk_chunk_size(n), L.hipk_chunk_count(n)) == oracle.chunk_geom(n)


def test_structs_match_header_layout():
    from pytorch_sparse_solver import _hipk
    assert ctypes.sizeof(_hipk.Params) == 48 and ctypes.sizeof(_hipk.Stats) == 96


def test_cuda_tensor_path_fails_loudly_without_library(monkeypatch):
    """No silent fallback: a missing extension is an error on the HIP path."""
    from pytorch_sparse_solver import _hipk
    monkeypatch.setattr(_hipk, "_lib", None)
    monkeypatch.setattr(_hipk, "LIB_PATH", "/nonexistent/libhipk.so")
    with pytest.raises(_hipk.HipkError, match="no CPU fallback"):
        _hipk.lib()


def test_cache_key_distinguishes_views():
    """a dense matrix and its transposed VIEW share storage, version and numel -- the handle cache key
    must still differ (the adjoint solve of the implicit-diff backward asks for the handle of A.T right after A's)."""
    import torch
    from pytorch_sparse_solver import _hipk
    A = torch.arange(16, dtype=torch.float64).reshape(4, 4)
    assert _hipk._cache_key(A) != _hipk._cache_key(A.T)
    assert _hipk._cache_key(A) == _hipk._cache_key(A.detach())
    assert _hipk._cache_key(A) != _hipk._cache_key(A[::1, :].clone())
    B = torch.arange(32, dtype=torch.float64).reshape(8, 4)
    assert _hipk._cache_key(B[:4]) != _hipk._cache_key(B[4:])            # same storage, different offset
    C = torch.randn(4, 4, dtype=torch.complex128)
    assert _hipk._cache_key(C) != _hipk._cache_key(C.conj())             # lazy conj bit
    S = A.to_sparse_csr()
    assert _hipk._cache_key(S) == _hipk._cache_key(S.detach())
    assert _hipk._cache_key(S) != _hipk._cache_key(S.t())                # CSC view of the same arrays


def test_gmres_large_restart_warns_once_about_the_route(monkeypatch):
    """gmres(restart > 255) on a device matrix leaves the HIP path (up to 255 it stays on the kernels since round 3): that must not be
    silent."""
    import warnings
    import torch
    from pytorch_sparse_solver.module_a import torch_sparse_linalg as T
    monkeypatch.setattr(T, "_fast_ok", lambda A, b, x0, M: True)         # pretend A, b live on the GPU
    monkeypatch.setattr(T, "_warned_restart", False)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        T._warn_restart_route(None, None, None, 300)
        T._warn_restart_route(None, None, None, 300)
        T._warn_restart_route(None, None, None, 40)
        T._warn_restart_route(None, None, None, 20)
    assert len(w) == 1 and "restart > 255" in str(w[0].message)
