"""Unified dispatcher (reference: solver.py:84-576) with Module A wired to the MI355X path.

`SparseSolver.solve(A, b, method=..., backend='module_a')` keeps the reference's argument
handling and `SolverResult` record (solver.py:256-379).  Module B (pyamgx) and Module C
(cuDSS) are NVIDIA-only backends outside this build's scope: they are reported
unavailable and requesting them raises the same ValueError the reference raises on a
machine without them (solver.py:219-225).
"""
from dataclasses import dataclass
from enum import Enum
from typing import Callable, Dict, List, Optional, Tuple, Union

import torch

from .utils.availability import get_available_backends


class SolverMethod(Enum):
    CG = "cg"
    BICGSTAB = "bicgstab"
    GMRES = "gmres"
    AMG = "amg"
    DIRECT = "direct"


class SolverBackend(Enum):
    MODULE_A = "module_a"
    MODULE_B = "module_b"
    MODULE_C = "module_c"
    AUTO = "auto"


@dataclass
class SolverResult:
    x: torch.Tensor
    converged: bool
    iterations: Optional[int]
    residual: Optional[float]
    backend: str
    method: str


class SparseSolver:
    """Solve A x = b through a named backend; only 'module_a' (and 'auto') exist here."""

    def __init__(self, default_backend: str = "auto", default_method: str = "cg", verbose: bool = False):
        self.verbose = verbose
        self.default_backend = default_backend
        self.default_method = default_method
        self._available: Optional[Dict[str, bool]] = None
        self._module_a = None

    @property
    def available_backends(self) -> List[str]:
        if self._available is None:
            self._check_backends()
        return [k for k, v in self._available.items() if v]

    def _check_backends(self) -> None:
        self._available = get_available_backends()
        if self.verbose:
            for name, ok in self._available.items():
                print(f"  {'[x]' if ok else '[ ]'} {name}")

    def _load_module_a(self):
        if self._module_a is None:
            try:
                from .module_a import bicgstab, cg, gmres
            except ImportError as e:  # pragma: no cover
                raise RuntimeError(f"Failed to load Module A: {e}")
            self._module_a = {'cg': cg, 'bicgstab': bicgstab, 'gmres': gmres}
        return self._module_a

    def _select_backend(self, backend: str, method: str, A: torch.Tensor) -> Tuple[str, str]:
        available = self.available_backends
        if not available:
            raise RuntimeError("No sparse solver backends are available!")
        if backend != "auto":
            if backend not in available:
                raise ValueError(f"Backend '{backend}' is not available. Available backends: {available}")
            return backend, method
        if method == "direct":
            raise ValueError("Direct solver requires Module C (cuDSS), which is not available. "
                             "Use an iterative method (cg, bicgstab, gmres) instead.")
        if method == "amg":
            raise ValueError("AMG solver requires Module B (AMGX), which is not available.")
        return "module_a", method

    def solve(self, A: Union[torch.Tensor, Callable[[torch.Tensor], torch.Tensor]], b: torch.Tensor,
              x0: Optional[torch.Tensor] = None, method: Optional[str] = None, backend: Optional[str] = None,
              tol: float = 1e-5, atol: float = 0.0, maxiter: Optional[int] = None,
              M: Optional[Callable[[torch.Tensor], torch.Tensor]] = None, **kwargs) -> Tuple[torch.Tensor, SolverResult]:
        method = self.default_method if method is None else method
        backend = self.default_backend if backend is None else backend
        probe = A if isinstance(A, torch.Tensor) else b
        selected_backend, selected_method = self._select_backend(backend, method, probe)
        if self.verbose:
            print(f"Using backend: {selected_backend}, method: {selected_method}")
        if selected_backend == "module_a":
            return self._solve_module_a(A, b, x0, selected_method, tol, atol, maxiter, M, **kwargs)
        raise ValueError(f"Unknown backend: {selected_backend}")

    def _solve_module_a(self, A, b, x0, method, tol, atol, maxiter, M, **kwargs) -> Tuple[torch.Tensor, SolverResult]:
        """solver.py:320-379: marshal kwargs, solve, recompute the relative residual."""
        module = self._load_module_a()
        if method not in module:
            raise ValueError(f"Method '{method}' not available in Module A. Use: {list(module.keys())}")
        call = {'tol': tol, 'atol': atol}
        if maxiter is not None:
            call['maxiter'] = maxiter
        if M is not None:
            call['M'] = M
        if x0 is not None:
            call['x0'] = x0
        if method == 'gmres':  # only these two extras are forwarded, and only to gmres
            for key in ('restart', 'solve_method'):
                if key in kwargs:
                    call[key] = kwargs[key]
        x, info = module[method](A, b, **call)
        residual = self._relative_residual(A, x, b)
        result = SolverResult(x=x, converged=(info == 0), iterations=None, residual=residual, backend="module_a",
                              method=method)
        return x, result

    @staticmethod
    def _relative_residual(A, x, b) -> float:
        if getattr(A, '_hipk_row_block', False):   # RowBlockCSR: ||b - A x|| / ||b|| of the GLOBAL system, from the solve's epilogue
            return A.relative_residual()
        if callable(A):
            Ax = A(x)
        elif A.is_cuda and A.layout in (torch.sparse_csr, torch.strided, torch.sparse_coo) and x.ndim == 1 \
                and not torch.is_complex(A) and A.dtype == x.dtype and A.dtype in (torch.float64, torch.float32):
            from . import _hipk
            Ax = _hipk.spmv(_hipk.handle_for(A), x.detach().contiguous())
        elif A.is_sparse:
            Ax = torch.sparse.mm(A, x.unsqueeze(-1)).squeeze(-1)
        else:
            Ax = torch.mv(A, x) if A.layout == torch.strided else torch.matmul(A, x)
        return torch.norm(b - Ax).item() / torch.norm(b).item()

    def cg(self, A, b, **kwargs):
        return self.solve(A, b, method='cg', **kwargs)

    def bicgstab(self, A, b, **kwargs):
        return self.solve(A, b, method='bicgstab', **kwargs)

    def gmres(self, A, b, **kwargs):
        return self.solve(A, b, method='gmres', **kwargs)

    def amg(self, A, b, **kwargs):
        return self.solve(A, b, method='amg', backend='module_b', **kwargs)

    def direct(self, A, b, **kwargs):
        return self.solve(A, b, method='direct', backend='module_c', **kwargs)

    def __repr__(self) -> str:
        return (f"SparseSolver(\n  available_backends={self.available_backends},\n"
                f"  default_backend='{self.default_backend}',\n  default_method='{self.default_method}'\n)")


_default_solver: Optional[SparseSolver] = None


def _get_default_solver() -> SparseSolver:
    global _default_solver
    if _default_solver is None:
        _default_solver = SparseSolver()
    return _default_solver


def solve(A, b, method: str = "cg", backend: str = "auto", **kwargs) -> Tuple[torch.Tensor, SolverResult]:
    return _get_default_solver().solve(A, b, method=method, backend=backend, **kwargs)


def cg(A, b, **kwargs):
    return solve(A, b, method='cg', **kwargs)


def bicgstab(A, b, **kwargs):
    return solve(A, b, method='bicgstab', **kwargs)


def gmres(A, b, **kwargs):
    return solve(A, b, method='gmres', **kwargs)


def amg(A, b, **kwargs):
    return solve(A, b, method='amg', backend='module_b', **kwargs)


def direct_solve(A, b, **kwargs):
    return solve(A, b, method='direct', backend='module_c', **kwargs)
