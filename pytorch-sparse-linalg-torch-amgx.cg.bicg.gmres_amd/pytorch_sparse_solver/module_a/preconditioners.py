"""Preconditioners for the `M` argument of cg / bicgstab / gmres (TSL:1019-1021, 849).

The reference takes any callable `M`.  A `JacobiPreconditioner` IS such a callable (`M(v) = v / diag(A)`), so it
works on every path and with the reference's own solvers; on the HIP fast path `cg` recognises it and runs the
preconditioned iteration device-resident (`hipk_pcg_solve`: the scaling is fused into the update and direction
kernels, 16 n extra bytes per iteration instead of a separate pass) -- SURVEY 8f-3.
"""
import torch

__all__ = ["JacobiPreconditioner"]


def _diagonal(A: torch.Tensor) -> torch.Tensor:
    if A.layout == torch.strided:
        return torch.diagonal(A).clone()
    if A.layout == torch.sparse_csr:
        crow, col, val = A.crow_indices(), A.col_indices(), A.values()
        n = A.shape[0]
        rows = torch.repeat_interleave(torch.arange(n, device=val.device), crow[1:] - crow[:-1])
        on = col == rows
        d = torch.zeros(n, dtype=val.dtype, device=val.device)
        return d.index_add_(0, rows[on], val[on])          # duplicate diagonal entries add, as in A @ e_i
    if A.layout == torch.sparse_coo:
        Ac = A.coalesce()
        i, v = Ac.indices(), Ac.values()
        on = i[0] == i[1]
        d = torch.zeros(A.shape[0], dtype=v.dtype, device=v.device)
        return d.index_add_(0, i[0][on], v[on])
    return _diagonal(A.to_sparse_csr())


class JacobiPreconditioner:
    """M(v) = v / diag(A).  `dinv` is the reciprocal diagonal, computed once (one rounding per entry)."""

    def __init__(self, A: torch.Tensor):
        if not (isinstance(A, torch.Tensor) and A.ndim == 2 and A.shape[0] == A.shape[1]):
            raise ValueError("JacobiPreconditioner needs a square matrix tensor")
        d = _diagonal(A.detach())
        if bool((d == 0).any()):
            raise ValueError("JacobiPreconditioner: zero on the diagonal")
        self.dinv = torch.reciprocal(d)
        self.shape = tuple(A.shape)

    def __call__(self, v):
        return self.dinv.to(v.dtype) * v
