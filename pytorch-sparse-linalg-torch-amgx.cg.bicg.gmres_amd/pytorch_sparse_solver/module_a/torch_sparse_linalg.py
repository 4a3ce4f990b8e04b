"""Module A on MI355X: cg / bicgstab / gmres with the reference's call surface.

Drop-in for the reference's `module_a/torch_sparse_linalg.py` (= TSL below; signatures
TSL:1019-1021, 1091-1093, 641-644; `(x, info)` returns with info in {0, -1}).

Two execution paths behind the same functions:

* FAST PATH (the product): `A` is a real CUDA/ROCm tensor (CSR, COO or dense -- converted
  to CSR once and cached), `b`/`x0` are single tensors on that device, `M is None`.
  The whole solve runs device-resident in libhipk.so (hand-written gfx950 kernels,
  include/hipk.h); Python only marshals pointers.  If the library is missing this path
  RAISES -- there is no silent fallback for CUDA tensors.
* GENERIC PATH: callable `A`, preconditioner `M`, PyTree/complex operands or CPU tensors.
  Same algorithms written once over flat vectors in torch ops (the reference walks
  PyTrees in every vector operation, TSL:165-203; here a PyTree is ravelled once).

Behaviour kept from the reference: fp64/complex128 promotion of b and x0 (TSL:979-980,
716-717), `maxiter = 10 n` default (TSL:982-984, 719-721), python-float tolerances
entering as fp32 tensors (TSL:816, 1010), the device/size dependent GMRES tolerance
(TSL:735-753) and its 10x slack in `info` (TSL:769-771), `info` decided by the TRUE
residual after the loop (TSL:1007-1014), BiCGStab breakdown rules (TSL:902-936), error
types/messages (TSL:181-183, 207-208, 727, 996, 1000-1002, 760).  What the reference
never returns (iteration counts) is available from `get_last_stats()`.
"""
from __future__ import annotations

import math
import os
from typing import Any, Callable, Optional, Tuple, Union

import torch

from .torch_tree_util import Partial, tree_leaves, tree_map, tree_ravel

DEFAULT_DTYPE = torch.float64
DEFAULT_COMPLEX_DTYPE = torch.complex128
_INV_SQRT2 = 0.7071067811865476

_last_stats = None


def get_last_stats():
    """Statistics of the most recent solve in this process (iterations, matvecs, breakdown code,
    true residual, device time).  Side channel only: the `(x, info)` return is unchanged."""
    return _last_stats


def _set_stats(st) -> None:
    global _last_stats
    _last_stats = st


@Partial
def _identity(x: Any) -> Any:
    return x


# ------------------------------------------------------------------------- operators
def _normalize_matvec(f):
    """Tensor or callable -> callable on the user's (PyTree) vectors (TSL:176-208)."""
    if callable(f):
        return f
    if isinstance(f, torch.Tensor):
        if f.ndim != 2 or f.shape[0] != f.shape[1]:
            raise ValueError(f'linear operator must be a square matrix, but has shape: {f.shape}')

        def matrix_mv(v_tree):
            flat, unravel = tree_ravel(v_tree)
            return unravel(torch.matmul(f, flat))

        return matrix_mv
    raise TypeError(f'linear operator must be either a function or tensor: {f}')


def _promote(x: torch.Tensor) -> torch.Tensor:
    return x.to(DEFAULT_COMPLEX_DTYPE) if torch.is_complex(x) else x.to(DEFAULT_DTYPE)


def _rdot(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """Real part of <a, b> without the cross terms (`_vdot_real_part`, TSL:100-127)."""
    if torch.is_complex(a) or torch.is_complex(b):
        a = a.to(DEFAULT_COMPLEX_DTYPE)
        b = b.to(DEFAULT_COMPLEX_DTYPE)
        return torch.vdot(a.real.contiguous(), b.real.contiguous()) + torch.vdot(a.imag.contiguous(),
                                                                                  b.imag.contiguous())
    return torch.vdot(a, b)


def _nrm(a: torch.Tensor) -> torch.Tensor:
    return torch.sqrt(torch.clamp(_rdot(a, a), min=0.0))


class _Flat:
    """The problem reduced to flat vectors: b, x0 (1-D, promoted) and flat operators."""

    def __init__(self, A, b, x0, M):
        if x0 is None:
            x0 = tree_map(torch.zeros_like, b)
        b = tree_map(_promote, b)
        x0 = tree_map(_promote, x0)
        b_leaves, x0_leaves = tree_leaves(b), tree_leaves(x0)
        if len(b_leaves) != len(x0_leaves):
            raise ValueError('x0 and b must have matching tree structure')
        self.ident = M is None or M is _identity
        A_fn = _normalize_matvec(A)
        M_fn = _identity if M is None else _normalize_matvec(M)
        self._shape_pairs = [(bl.shape, xl.shape) for bl, xl in zip(b_leaves, x0_leaves)]
        self.b, self.unravel = tree_ravel(b)
        self.x0, _ = tree_ravel(x0)
        self.size = self.b.numel()
        unravel = self.unravel
        self.A = lambda v: tree_ravel(A_fn(unravel(v)))[0]
        self.M = (lambda v: v) if self.ident else (lambda v: tree_ravel(M_fn(unravel(v)))[0])

    def check_shapes(self):
        for bs, xs in self._shape_pairs:
            if bs != xs:
                raise ValueError(f'arrays in x0 and b must have matching shapes: {xs} vs {bs}')


# ------------------------------------------------------------------------- fast path
def _jacobi_of(M):
    """The JacobiPreconditioner behind `M`, if that is what it is (the fast path runs it device-resident)."""
    from .preconditioners import JacobiPreconditioner
    return M if isinstance(M, JacobiPreconditioner) else None


def _fast_ok(A, b, x0, M) -> bool:
    return (isinstance(A, torch.Tensor) and A.is_cuda and A.ndim == 2 and not torch.is_complex(A)
            and A.dtype in (torch.float64, torch.float32)
            and isinstance(b, torch.Tensor) and b.is_cuda and b.device == A.device and b.ndim == 1
            and not torch.is_complex(b)
            and (x0 is None or (isinstance(x0, torch.Tensor) and x0.device == b.device and not torch.is_complex(x0)))
            and (M is None or M is _identity))


def _fast_solve(method: str, A, b, x0, tol, atol, maxiter, restart=20, solve_method='batched', jacobi=None):
    from .. import _hipk

    if A.shape[0] != A.shape[1]:
        raise ValueError(f'linear operator must be a square matrix, but has shape: {A.shape}')
    if x0 is not None and x0.shape != b.shape:
        raise ValueError(f'arrays in x0 and b must have matching shapes: {x0.shape} vs {b.shape}')
    if A.shape[1] != b.numel():
        raise RuntimeError(f'size mismatch, got input ({A.shape[0]}x{A.shape[1]}), vec ({b.numel()})')
    if method == 'gmres' and solve_method not in ('batched', 'incremental'):
        raise ValueError(f"Unsupported solve_method: {solve_method}")
    h = _hipk.handle_for(A)  # raises HipkError when libhipk.so is missing: no fallback
    # fp64 is forced exactly like the reference (TSL:979-980). fp32 storage is an
    # extension, taken only when A itself is fp32 (the reference raises there).
    work_dtype = torch.float64 if h.dtype == torch.float64 else torch.float32
    bb = b.detach().to(work_dtype).contiguous()
    x = torch.zeros_like(bb) if x0 is None else x0.detach().to(work_dtype).clone().contiguous()
    if jacobi is not None:  # cg / gmres with M = diag(dinv): hipk_pcg_solve / hipk_pgmres_solve
        if jacobi.shape != tuple(A.shape):
            raise ValueError(f'preconditioner shape {jacobi.shape} does not match the operator {tuple(A.shape)}')
        dinv = jacobi.dinv.detach().to(device=bb.device, dtype=work_dtype).contiguous()
        if method == 'gmres':
            st = _hipk.solve_pgmres(h, dinv, bb, x, tol=tol, atol=atol, maxiter=maxiter, restart=restart,
                                    solve_method=solve_method)
        else:
            st = _hipk.solve_pcg(h, dinv, bb, x, tol=tol, atol=atol, maxiter=maxiter, method=method)
    else:
        st = _hipk.solve(method, h, bb, x, tol=tol, atol=atol, maxiter=maxiter, restart=restart,
                         solve_method=solve_method)
    _set_stats(st)
    return x, int(st.info)


def _device_vectors(b, x0) -> bool:
    return (isinstance(b, torch.Tensor) and b.is_cuda and b.ndim == 1 and not torch.is_complex(b)
            and b.dtype in (torch.float64, torch.float32)
            and (x0 is None or (isinstance(x0, torch.Tensor) and x0.device == b.device and x0.ndim == 1
                                and not torch.is_complex(x0))))


def _fast_solve_matrix_free(A, b, x0, tol, atol, maxiter, M):
    """cg() with a MATRIX-FREE operator (callable `A`) on device vectors: the fused vector kernels and the device-side
    stop word run the iteration, `A` (and `M`, if any) are called between them on the same stream
    (`_hipk.solve_cg_stepwise`).  fp64 like the reference (TSL:979-980)."""
    from .. import _hipk

    if x0 is not None and x0.shape != b.shape:
        raise ValueError(f'arrays in x0 and b must have matching shapes: {x0.shape} vs {b.shape}')
    bb = b.detach().to(torch.float64).contiguous()
    x = torch.zeros_like(bb) if x0 is None else x0.detach().to(torch.float64).clone().contiguous()
    M_fn = None if (M is None or M is _identity) else _normalize_matvec(M)
    st = _hipk.solve_cg_stepwise(None, _normalize_matvec(A), M_fn, bb, x, tol=tol, atol=atol, maxiter=maxiter)
    _set_stats(st)
    return x, int(st.info)


def _fast_solve_matrix_free_c(kind, A, b, x0, tol, atol, maxiter, M, restart=20, solve_method='batched'):
    """cg() / bicgstab() / gmres() with a MATRIX-FREE operator on device vectors, on the device-resident C loops
    (`_hipk.solve_matrix_free`: hipk_op_create + hipk_{cg,bicgstab,gmres}_solve and their preconditioned forms): no host
    synchronisation inside CG / BiCGStab, one per GMRES cycle.  fp64 like the reference (TSL:979-980)."""
    from .. import _hipk

    if x0 is not None and x0.shape != b.shape:
        raise ValueError(f'arrays in x0 and b must have matching shapes: {x0.shape} vs {b.shape}')
    if kind == 'gmres' and solve_method not in ('batched', 'incremental'):
        raise ValueError(f"Unsupported solve_method: {solve_method}")
    bb = b.detach().to(torch.float64).contiguous()
    x = torch.zeros_like(bb) if x0 is None else x0.detach().to(torch.float64).clone().contiguous()
    A_fn = _normalize_matvec(A)
    jac = _jacobi_of(M)
    if jac is not None:
        if jac.shape != (bb.numel(), bb.numel()):
            raise ValueError(f'preconditioner shape {jac.shape} does not match the operator {(bb.numel(), bb.numel())}')
        st = _hipk.solve_matrix_free(kind, A_fn, bb, x, tol=tol, atol=atol, maxiter=maxiter, restart=restart,
                                     solve_method=solve_method, dinv=jac.dinv.detach().to(device=bb.device, dtype=torch.float64).contiguous())
    else:
        M_fn = None if (M is None or M is _identity) else _normalize_matvec(M)
        st = _hipk.solve_matrix_free(kind, A_fn, bb, x, tol=tol, atol=atol, maxiter=maxiter, restart=restart,
                                     solve_method=solve_method, M=M_fn)
    _set_stats(st)
    return x, int(st.info)


def _fast_solve_callable(kind, A, b, x0, tol, atol, maxiter, M, restart=20, solve_method='batched'):
    """cg() / bicgstab() / gmres() with an arbitrary preconditioner `M` (callable or matrix) and a device CSR/dense `A`:
    the fused kernels run the iteration, `M` is called between them on the same stream (`_hipk.solve_cg_callable` through
    the step API, `_hipk.solve_bicgstab_callable` / `solve_gmres_callable` through the C loops' callback)."""
    from .. import _hipk

    if A.shape[0] != A.shape[1]:
        raise ValueError(f'linear operator must be a square matrix, but has shape: {A.shape}')
    if x0 is not None and x0.shape != b.shape:
        raise ValueError(f'arrays in x0 and b must have matching shapes: {x0.shape} vs {b.shape}')
    if A.shape[1] != b.numel():
        raise RuntimeError(f'size mismatch, got input ({A.shape[0]}x{A.shape[1]}), vec ({b.numel()})')
    h = _hipk.handle_for(A)
    work_dtype = torch.float64 if h.dtype == torch.float64 else torch.float32
    bb = b.detach().to(work_dtype).contiguous()
    x = torch.zeros_like(bb) if x0 is None else x0.detach().to(work_dtype).clone().contiguous()
    if kind == 'gmres':
        st = _hipk.solve_gmres_callable(h, _normalize_matvec(M), bb, x, tol=tol, atol=atol, maxiter=maxiter,
                                        restart=restart, solve_method=solve_method)
    else:
        run = _hipk.solve_cg_callable if kind == 'cg' else _hipk.solve_bicgstab_callable
        st = run(h, _normalize_matvec(M), bb, x, tol=tol, atol=atol, maxiter=maxiter)
    _set_stats(st)
    return x, int(st.info)


# ------------------------------------------------------------------------- generic algorithms
class _GenericStats:
    def __init__(self, method, iterations, matvecs, info, breakdown=0):
        self.method, self.iterations, self.matvecs, self.info, self.breakdown = method, iterations, matvecs, info, breakdown


def _sq_tol(tol, atol, bs):
    dev = bs.device
    return torch.maximum(torch.square(torch.tensor(tol, device=dev)) * bs, torch.square(torch.tensor(atol, device=dev)))


def _cg_flat(P: _Flat, tol, atol, maxiter):
    A, M, b = P.A, P.M, P.b
    dtype = b.dtype
    atol2 = _sq_tol(tol, atol, _rdot(b, b))
    x = P.x0
    r = b - A(x)
    z = M(r)
    p = z
    gamma = _rdot(r, z).to(dtype)
    k = 0
    while True:
        rs = (gamma.real if torch.is_complex(gamma) else gamma) if P.ident else _rdot(r, r)
        if k >= maxiter or bool(rs <= atol2):
            break
        Ap = A(p)
        alpha = gamma / _rdot(p, Ap).to(dtype)
        x = x + alpha * p
        r = r - alpha * Ap
        z = M(r)
        gamma_new = _rdot(r, z).to(dtype)
        p = z + (gamma_new / gamma) * p
        gamma = gamma_new
        k += 1
    return x, k, k + 1, 0


def _bicgstab_flat(P: _Flat, tol, atol, maxiter):
    A, M, b = P.A, P.M, P.b
    dtype, dev = b.dtype, b.device
    eps = torch.finfo(dtype).eps
    atol2 = _sq_tol(tol, atol, _rdot(b, b))
    x = P.x0
    r = b - A(x)
    rhat = r
    one = torch.tensor(1.0, dtype=dtype, device=dev)
    alpha, omega, rho = one, one, one
    p = q = r
    k, code, mv = 0, 0, 1
    while k < maxiter:
        if bool(_rdot(r, r) <= atol2):
            break
        rho_new = torch.vdot(rhat, r)
        if bool(torch.abs(rho_new) < eps * torch.abs(rho)):
            code = -10
            break
        beta = rho_new / rho * alpha / omega
        p = r + beta * (p - omega * q)
        phat = M(p)
        q = A(phat)
        mv += 1
        alpha_new = rho_new / torch.vdot(rhat, q)
        if bool(torch.abs(alpha_new) < eps):
            code = -11
            break
        s = r - alpha_new * q
        exit_early = bool(_rdot(s, s) < atol2)
        shat = M(s)
        t = A(shat)
        mv += 1
        tt = torch.vdot(t, t)
        omega_new = torch.zeros((), dtype=dtype, device=dev) if bool(torch.abs(tt) < eps) else torch.vdot(t, s) / tt
        if bool(torch.abs(omega_new) < eps) and not exit_early:
            code = -11
            break
        if exit_early:
            x = x + alpha_new * phat
            r = s
        else:
            x = x + (alpha_new * phat + omega_new * shat)
            r = s - omega_new * t
        rho, alpha, omega = rho_new, alpha_new, omega_new
        k += 1
        if exit_early:
            break
    return x, k, mv, code


def _safe_normalize(v: torch.Tensor, thresh=None):
    """(v/||v||, ||v||), or (0, 0) when ||v|| <= thresh (default eps) -- TSL:217-273."""
    norm = _nrm(v)
    if thresh is None:
        thresh = torch.finfo(norm.dtype).eps
    use = norm > thresh
    unit = torch.where(use, v / norm.to(v.dtype), torch.zeros_like(v))
    return unit, torch.where(use, norm, torch.zeros_like(norm))


def _givens(a: torch.Tensor, b: torch.Tensor):
    """Rotation (cs, sn) zeroing b against a (TSL:508-518)."""
    if bool(torch.abs(b) == 0):
        return torch.ones_like(a), torch.zeros_like(a)
    if bool(torch.abs(a) < torch.abs(b)):
        t = -a / b
        r = torch.rsqrt(1 + torch.abs(t) ** 2).to(t.dtype)
        return r * t, r
    t = -b / a
    r = torch.rsqrt(1 + torch.abs(t) ** 2).to(t.dtype)
    return r, r * t


def _normal_eq_lstsq(H: torch.Tensor, rhs: torch.Tensor) -> torch.Tensor:
    """argmin ||H y - rhs|| through the normal equations + Cholesky (TSL:407-421)."""
    Hh = H.conj().T
    a2, b2 = Hh @ H, (Hh @ rhs).unsqueeze(-1)
    try:
        sol = torch.cholesky_solve(b2, torch.linalg.cholesky(a2))
    except RuntimeError:
        sol = torch.linalg.solve(a2, b2)
    return sol.squeeze(-1)


def _gmres_flat(P: _Flat, atol_eff, ptol, restart, maxiter, incremental: bool):
    A, M, b = P.A, P.M, P.b
    dtype, dev, n, m = b.dtype, b.device, b.numel(), restart
    rdtype = torch.float64
    eps = torch.finfo(rdtype).eps
    x = P.x0
    unit, rnorm = _safe_normalize(M(b - A(x)))
    cycles, mv, happy = 0, 1, 0
    V = torch.empty(m + 1, n, dtype=dtype, device=dev)  # basis vectors are ROWS: contiguous; allocated once per solve
    H = torch.empty(m + 1, m, dtype=dtype, device=dev)
    while cycles < maxiter and bool(rnorm > atol_eff):
        V.zero_()
        V[0] = unit
        H.zero_()
        if incremental:
            R = torch.eye(m, dtype=dtype, device=dev)
            rot = []
            beta_vec = torch.zeros(m + 1, dtype=dtype, device=dev)
            beta_vec[0] = rnorm.to(dtype)
        k, breakdown, err = 0, False, rnorm
        while k < m and not breakdown and (not incremental or bool(err > ptol)):
            w = M(A(V[k]))
            mv += 1
            _, n0 = _safe_normalize(w)
            Vk = V[:k + 1]
            hsum = torch.zeros(k + 1, dtype=dtype, device=dev)
            q = w
            for cgs_pass in range(2):
                if cgs_pass == 1:
                    _, rn = _safe_normalize(hsum)
                    if not bool(rn < qn * _INV_SQRT2):
                        break
                h = Vk.conj() @ q
                q = q - Vk.T @ h
                hsum = hsum + h
                _, qn = _safe_normalize(q)
            unit_v, n1 = _safe_normalize(q, thresh=eps * n0)
            V[k + 1] = unit_v
            H[:k + 1, k] = hsum
            H[k + 1, k] = n1.to(dtype)
            breakdown = bool(n1 == 0)
            if incremental:
                col = H[:k + 2, k].clone()
                for i, (cs, sn) in enumerate(rot):
                    hi = cs.conj() * col[i] - sn.conj() * col[i + 1]
                    col[i + 1] = sn * col[i] + cs * col[i + 1]
                    col[i] = hi
                cs, sn = _givens(col[k], col[k + 1])
                rot.append((cs, sn))
                col[k] = cs.conj() * col[k] - sn.conj() * col[k + 1]
                R[:k + 1, k] = col[:k + 1]
                bk = cs.conj() * beta_vec[k] - sn.conj() * beta_vec[k + 1]
                beta_vec[k + 1] = sn * beta_vec[k] + cs * beta_vec[k + 1]
                beta_vec[k] = bk
                err = torch.abs(beta_vec[k + 1])
            k += 1
        if breakdown:
            happy = 1
        if k > 0:
            if incremental:
                y = torch.linalg.solve_triangular(R[:k, :k], beta_vec[:k].unsqueeze(-1), upper=True).squeeze(-1)
            else:
                rhs = torch.zeros(k + 1, dtype=dtype, device=dev)
                rhs[0] = rnorm.to(dtype)
                y = _normal_eq_lstsq(H[:k + 1, :k], rhs)
            x = x + V[:k].T @ y
        unit, rnorm = _safe_normalize(M(b - A(x)))
        mv += 1
        cycles += 1
    return x, cycles, mv, happy


# ------------------------------------------------------------------------- public wrappers
def _flat_tree_operands(A, b, x0):
    """A device MATRIX with PyTree right-hand sides (`b` a dict / tuple / list of device tensors -- or a tensor that is not
    1-D): `_normalize_matvec` applies a matrix to the concatenation of the leaves (TSL:186-205), so the solve is the flat one.
    Returns (b_flat, x0_flat, unravel) for the fast path, or None when the operands are not of that kind.  Same structure
    errors as the generic path."""
    if not (isinstance(A, torch.Tensor) and A.is_cuda and A.ndim == 2 and not torch.is_complex(A)):
        return None
    if isinstance(b, torch.Tensor) and b.ndim == 1:
        return None
    leaves = tree_leaves(b)
    if not leaves or not all(isinstance(v, torch.Tensor) and v.device == A.device and not torch.is_complex(v) for v in leaves):
        return None
    if x0 is not None:
        xl = tree_leaves(x0)
        if len(xl) != len(leaves):
            raise ValueError('x0 and b must have matching tree structure')
        for bv, xv in zip(leaves, xl):
            if not isinstance(xv, torch.Tensor) or xv.device != A.device or torch.is_complex(xv):
                return None
            if bv.shape != xv.shape:
                raise ValueError(f'arrays in x0 and b must have matching shapes: {xv.shape} vs {bv.shape}')
    bf, unravel = tree_ravel(b)
    return bf, (None if x0 is None else tree_ravel(x0)[0]), unravel


def _isolve(kind: str, A, b, x0, tol, atol, maxiter, M):
    """Shared CG/BiCGStab wrapper (`_isolve`, TSL:968-1016)."""
    flat = _flat_tree_operands(A, b, x0)
    if flat is not None and (M is None or M is _identity or _jacobi_of(M) is not None):
        x, info = _isolve(kind, A, flat[0], flat[1], tol, atol, maxiter, M)     # the flat solve on the fast path
        return flat[2](x), info
    if _fast_ok(A, b, x0, M):
        return _fast_solve(kind, A, b, x0, tol, atol, maxiter)
    if _jacobi_of(M) is not None and _fast_ok(A, b, x0, None):
        return _fast_solve(kind, A, b, x0, tol, atol, maxiter, jacobi=_jacobi_of(M))
    if M is not None and _fast_ok(A, b, x0, None) and os.environ.get('HIPK_CG_CALLABLE_M', '1') != '0':
        return _fast_solve_callable(kind, A, b, x0, tol, atol, maxiter, M)   # any other M: fused kernels around the callable
    if (callable(A) and not isinstance(A, torch.Tensor) and _device_vectors(b, x0)
            and os.environ.get('HIPK_CG_MATRIX_FREE', '1') != '0'):
        # matrix-free operator (TSL:176-208 takes a callable for every solver): the device-resident C loops call `A` where they
        # launch their SpMV (hipk_op_create); cg with a callable M keeps the host-driven step loop (HIPK_CG_MATRIX_FREE=step: always)
        if (kind == 'cg' and M is not None and M is not _identity and _jacobi_of(M) is None) \
                or (kind == 'cg' and os.environ.get('HIPK_CG_MATRIX_FREE') == 'step'):
            return _fast_solve_matrix_free(A, b, x0, tol, atol, maxiter, M)
        return _fast_solve_matrix_free_c(kind, A, b, x0, tol, atol, maxiter, M)
    P = _Flat(A, b, x0, M)
    if maxiter is None:
        maxiter = 10 * P.size
    P.check_shapes()
    body = _cg_flat if kind == 'cg' else _bicgstab_flat
    x, iters, mv, code = body(P, tol, atol, maxiter)
    final_residual = _nrm(P.M(P.b - P.A(x)))
    b_norm = _nrm(P.b)
    thr = torch.maximum(torch.tensor(tol, device=b_norm.device) * b_norm, torch.tensor(atol, device=b_norm.device))
    failed = bool(torch.isnan(_nrm(x))) or bool(final_residual > thr)
    info = -1 if failed else 0
    _set_stats(_GenericStats(kind, iters, mv + 1, info, code))
    return P.unravel(x), info


def _is_row_block(A) -> bool:
    """A `distributed.RowBlockCSR` operand: this rank's rows of a global system (one process per GPU)."""
    return getattr(A, '_hipk_row_block', False) is True


def _dist_solve(kind: str, A, b, x0, tol, atol, maxiter, M, restart=20, solve_method='batched'):
    """cg / bicgstab / gmres on a RowBlockCSR operand: the row-partitioned solvers (distributed.py, csrc/hipk_dist.hip) behind the
    reference's call surface.  Returns this rank's slice of x and the (rank-independent) info."""
    if M is not None and M is not _identity:
        raise ValueError(f"{kind}: preconditioners are not available on a RowBlockCSR (row-partitioned) operand")
    if kind == 'gmres' and solve_method not in ('batched', 'incremental'):
        raise ValueError(f"Unsupported solve_method: {solve_method}")
    x, info, st = A.solve(kind, b, x0, tol=tol, atol=atol, maxiter=maxiter, restart=restart, solve_method=solve_method)
    _set_stats(st)
    return x, info


def _use_implicit_diff(A: Any, b: Any) -> bool:
    return (isinstance(A, torch.Tensor) and isinstance(b, torch.Tensor) and A.ndim == 2
            and (A.requires_grad or b.requires_grad))


def _transpose_of(A: torch.Tensor) -> torch.Tensor:
    """A^H for the adjoint solve (TSL:1245). `.T` is not implemented for sparse-compressed CUDA tensors, so sparse
    layouts go through `.t()` (CSR -> CSC view of the same data; the handle builder converts it back to CSR once)."""
    At = A.T if A.layout == torch.strided else A.t()
    return At.conj() if torch.is_complex(A) else At


class ImplicitAdjointFunction(torch.autograd.Function):
    """Gives an already computed solution x = A^-1 b its implicit-function backward:
    grad_b = A^-T grad_x, obtained with one more solve (TSL:1227-1248)."""

    @staticmethod
    def forward(ctx, A_matrix, b, x, transpose_solve_fn, *solve_args):
        ctx.save_for_backward(A_matrix)
        ctx.transpose_solve_fn = transpose_solve_fn
        ctx.solve_args = solve_args
        return x

    @staticmethod
    def backward(ctx, grad_output):
        (A_matrix,) = ctx.saved_tensors
        grad_b = None
        if ctx.needs_input_grad[1]:
            grad_b, _ = ctx.transpose_solve_fn(_transpose_of(A_matrix.detach()), grad_output.detach(), *ctx.solve_args)
        return (None, grad_b, None, None) + (None,) * len(ctx.solve_args)


class LinearSolveFunction(torch.autograd.Function):
    """x = solve_fn(A, b, *args) with backward grad_b = transpose_solve_fn(A^T, grad_x, *args)
    (TSL:1161-1224). Gradient w.r.t. A is not produced, as in the reference."""

    @staticmethod
    def forward(ctx, A_matrix, b, solve_fn, transpose_solve_fn, *solve_args):
        x, _ = solve_fn(A_matrix.detach(), b.detach(), *solve_args)
        ctx.save_for_backward(A_matrix)
        ctx.transpose_solve_fn = transpose_solve_fn
        ctx.solve_args = solve_args
        return x

    @staticmethod
    def backward(ctx, grad_output):
        (A_matrix,) = ctx.saved_tensors
        grad_b = None
        if ctx.needs_input_grad[1]:
            grad_b, _ = ctx.transpose_solve_fn(_transpose_of(A_matrix.detach()), grad_output.detach(), *ctx.solve_args)
        return (None, grad_b, None, None) + (None,) * len(ctx.solve_args)


def cg(A: Union[torch.Tensor, Callable[[Any], Any]], b: Any, x0: Optional[Any] = None,
       *, tol: float = 1e-5, atol: float = 0.0, maxiter: Optional[int] = None,
       M: Optional[Callable[[Any], Any]] = None) -> Tuple[Any, Optional[int]]:
    """Conjugate gradients for hermitian positive definite `A` (TSL:1019-1088).

    Returns `(x, info)`; info = 0 iff ||b - A x|| <= max(tol ||b||, atol) for the returned x.
    `A` may be a `RowBlockCSR` (this rank's rows of a global system inside a process group): `b`, `x0`, `x` are then the rank's slices.
    """
    if _is_row_block(A):
        return _dist_solve('cg', A, b, x0, tol, atol, maxiter, M)
    diff = _use_implicit_diff(A, b)
    A_, b_ = (A.detach(), b.detach()) if diff else (A, b)
    x, info = _isolve('cg', A_, b_, x0, tol, atol, maxiter, M)
    if diff:
        def transpose_solve_fn(A_mat, rhs, x_init=None, tol_=1e-5, atol_=0.0, maxiter_=None):
            return _isolve('cg', A_mat, rhs, x_init, tol_, atol_, maxiter_, M)

        x = ImplicitAdjointFunction.apply(A, b, x, transpose_solve_fn, x0, tol, atol, maxiter)
    return x, info


def bicgstab(A: Union[torch.Tensor, Callable[[Any], Any]], b: Any, x0: Optional[Any] = None,
             *, tol: float = 1e-5, atol: float = 0.0, maxiter: Optional[int] = None,
             M: Optional[Callable[[Any], Any]] = None) -> Tuple[Any, Optional[int]]:
    """BiCGStab for general square `A` (TSL:1091-1154). Returns `(x, info)`.  `A` may be a `RowBlockCSR` (see `cg`)."""
    if _is_row_block(A):
        return _dist_solve('bicgstab', A, b, x0, tol, atol, maxiter, M)
    diff = _use_implicit_diff(A, b)
    A_, b_ = (A.detach(), b.detach()) if diff else (A, b)
    x, info = _isolve('bicgstab', A_, b_, x0, tol, atol, maxiter, M)
    if diff:
        def transpose_solve_fn(A_mat, rhs, x_init=None, tol_=1e-5, atol_=0.0, maxiter_=None):
            return _isolve('bicgstab', A_mat, rhs, x_init, tol_, atol_, maxiter_, M)

        x = ImplicitAdjointFunction.apply(A, b, x, transpose_solve_fn, x0, tol, atol, maxiter)
    return x, info


_warned_restart = False
# the device-resident GMRES keeps its Hessenberg arrays in the handle's header up to restart 31 and in a workspace block sized for
# the restart up to 255 (csrc/hipk_gmres.hip: HIPK_GM_MAXM_BIG); the reference accepts any restart (TSL:641-644)
_HIP_MAX_RESTART = 255


def _warn_restart_route(A, b, x0, restart) -> None:
    """gmres(restart > 255) on a device matrix leaves the HIP path (routing rule below): say so once, loudly --
    at N = 4 M the generic torch-op path is orders of magnitude slower."""
    global _warned_restart
    if not _warned_restart and restart > _HIP_MAX_RESTART and _fast_ok(A, b, x0, None):
        import warnings
        _warned_restart = True
        warnings.warn(f"gmres(restart={restart}): the device-resident HIP GMRES keeps at most {_HIP_MAX_RESTART} basis vectors; "
                      f"restart > {_HIP_MAX_RESTART} runs on the generic torch-op path (one host synchronisation per Arnoldi step, "
                      f"much slower on large systems). Use restart <= {_HIP_MAX_RESTART} to stay on the MI355X kernels.",
                      RuntimeWarning, stacklevel=4)


def _gmres_impl(A, b, x0, tol, atol, restart, maxiter, M, solve_method):
    # the device-resident GMRES keeps at most 255 basis vectors (restart <= 31: H in the header block and the small-system
    # one-launch kernels; beyond: H in a workspace block, launch sequences); larger Krylov
    # spaces take the generic path (documented routing rule, not a fallback on failure; warned about once)
    flat = _flat_tree_operands(A, b, x0)
    if flat is not None and (M is None or M is _identity or _jacobi_of(M) is not None):
        x, info = _gmres_impl(A, flat[0], flat[1], tol, atol, restart, maxiter, M, solve_method)
        return flat[2](x), info
    _warn_restart_route(A, b, x0, restart)
    if _fast_ok(A, b, x0, M) and 1 <= restart <= _HIP_MAX_RESTART:
        return _fast_solve('gmres', A, b, x0, tol, atol, maxiter, restart=restart, solve_method=solve_method)
    if _jacobi_of(M) is not None and _fast_ok(A, b, x0, None) and 1 <= restart <= _HIP_MAX_RESTART:
        return _fast_solve('gmres', A, b, x0, tol, atol, maxiter, restart=restart, solve_method=solve_method,
                           jacobi=_jacobi_of(M))
    if (M is not None and _fast_ok(A, b, x0, None) and 1 <= restart <= _HIP_MAX_RESTART
            and os.environ.get('HIPK_CG_CALLABLE_M', '1') != '0'):
        if solve_method not in ('batched', 'incremental'):
            raise ValueError(f"Unsupported solve_method: {solve_method}")
        return _fast_solve_callable('gmres', A, b, x0, tol, atol, maxiter, M, restart=restart, solve_method=solve_method)
    if (callable(A) and not isinstance(A, torch.Tensor) and _device_vectors(b, x0) and 1 <= restart <= _HIP_MAX_RESTART
            and os.environ.get('HIPK_CG_MATRIX_FREE', '1') != '0'):
        return _fast_solve_matrix_free_c('gmres', A, b, x0, tol, atol, maxiter, M, restart=restart, solve_method=solve_method)
    P = _Flat(A, b, x0, M)
    if maxiter is None:
        maxiter = 10 * P.size
    b_norm = _nrm(P.b)
    dev = P.b.device
    size = P.size
    real_dtype = torch.float64
    # TSL:735-748: size- and device-dependent effective tolerance
    if dev.type == 'cuda':
        adaptive_tol = max(tol, 1e-12 * torch.sqrt(torch.tensor(size, dtype=torch.float64)))
        base_atol = torch.finfo(real_dtype).eps * 1000 * size
    else:
        adaptive_tol = max(tol, 1e-14 * torch.sqrt(torch.tensor(size, dtype=torch.float64)))
        base_atol = torch.finfo(real_dtype).eps * 100 * size
    adaptive_t = adaptive_tol.to(dev) if isinstance(adaptive_tol, torch.Tensor) else torch.tensor(adaptive_tol, device=dev)
    atol_eff = torch.maximum(adaptive_t * b_norm,
                             torch.maximum(torch.tensor(atol, device=dev), torch.tensor(base_atol, device=dev)))
    ptol = _nrm(P.M(P.b)) * torch.minimum(torch.tensor(1.0, device=dev), atol_eff / b_norm)
    if solve_method not in ('incremental', 'batched'):
        raise ValueError(f"Unsupported solve_method: {solve_method}")
    x, cycles, mv, happy = _gmres_flat(P, atol_eff, ptol, restart, maxiter, solve_method == 'incremental')
    final_residual = _nrm(P.M(P.b - P.A(x)))
    failed = bool(torch.isnan(_nrm(x))) or bool(final_residual > atol_eff * 10)
    info = -1 if failed else 0
    _set_stats(_GenericStats('gmres', cycles, mv + 1, info, happy))
    return P.unravel(x), info


def gmres(A: Union[torch.Tensor, Callable[[Any], Any]], b: Any, x0: Optional[Any] = None,
          *, tol: float = 1e-5, atol: float = 0.0, restart: int = 20,
          maxiter: Optional[int] = None, M: Optional[Callable[[Any], Any]] = None,
          solve_method: str = 'batched') -> Tuple[Any, Optional[int]]:
    """Restarted GMRES (TSL:641-784). `maxiter` counts restart cycles; `solve_method` is
    'batched' (least squares by normal equations at the end of a cycle) or 'incremental'
    (Givens QR with early exit inside a cycle). Returns `(x, info)`.  `A` may be a `RowBlockCSR` (see `cg`; restart <= 31 there)."""
    if _is_row_block(A):
        return _dist_solve('gmres', A, b, x0, tol, atol, maxiter, M, restart=restart, solve_method=solve_method)
    diff = _use_implicit_diff(A, b)
    A_, b_ = (A.detach(), b.detach()) if diff else (A, b)
    x, info = _gmres_impl(A_, b_, x0, tol, atol, restart, maxiter, M, solve_method)
    if diff:
        def transpose_solve_fn(A_mat, rhs, x_init=None, tol_=1e-5, atol_=0.0, restart_=20, maxiter_=None):
            return _gmres_impl(A_mat, rhs, x_init, tol_, atol_, restart_, maxiter_, M, solve_method)

        x = ImplicitAdjointFunction.apply(A, b, x, transpose_solve_fn, x0, tol, atol, restart, maxiter)
    return x, info


def _differentiable(name: str, solver, A, b, x0, tol, atol, maxiter, restart=None):
    if not isinstance(A, torch.Tensor) or A.ndim != 2:
        raise ValueError(f"For differentiable {name}, A must be a 2D tensor")
    if restart is None:
        def solve_fn(A_mat, rhs, x_init=None, tol_=1e-5, atol_=0.0, maxiter_=None):
            return solver(A_mat, rhs, x0=x_init, tol=tol_, atol=atol_, maxiter=maxiter_)

        return LinearSolveFunction.apply(A, b, solve_fn, solve_fn, x0, tol, atol, maxiter)

    def solve_fn_r(A_mat, rhs, x_init=None, tol_=1e-5, atol_=0.0, restart_=20, maxiter_=None):
        return solver(A_mat, rhs, x0=x_init, tol=tol_, atol=atol_, restart=restart_, maxiter=maxiter_)

    return LinearSolveFunction.apply(A, b, solve_fn_r, solve_fn_r, x0, tol, atol, restart, maxiter)


def cg_differentiable(A: torch.Tensor, b: torch.Tensor, x0: Optional[torch.Tensor] = None,
                      *, tol: float = 1e-5, atol: float = 0.0, maxiter: Optional[int] = None) -> torch.Tensor:
    """CG returning only x, differentiable w.r.t. b by an adjoint solve (TSL:1261-1294)."""
    return _differentiable("CG", cg, A, b, x0, tol, atol, maxiter)


def bicgstab_differentiable(A: torch.Tensor, b: torch.Tensor, x0: Optional[torch.Tensor] = None,
                            *, tol: float = 1e-5, atol: float = 0.0, maxiter: Optional[int] = None) -> torch.Tensor:
    """BiCGStab returning only x, differentiable w.r.t. b (TSL:1297-1330)."""
    return _differentiable("BiCGStab", bicgstab, A, b, x0, tol, atol, maxiter)


def gmres_differentiable(A: torch.Tensor, b: torch.Tensor, x0: Optional[torch.Tensor] = None,
                         *, tol: float = 1e-5, atol: float = 0.0, restart: int = 20,
                         maxiter: Optional[int] = None) -> torch.Tensor:
    """GMRES returning only x, differentiable w.r.t. b (TSL:1333-1367)."""
    return _differentiable("GMRES", gmres, A, b, x0, tol, atol, maxiter, restart=restart)
