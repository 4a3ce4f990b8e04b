"""Preconditioners for the `M` argument of cg / bicgstab / gmres (TSL:1019-1021, 849).

The reference takes any callable `M`.  A `JacobiPreconditioner` IS such a callable (`M(v) = v / diag(A)`), so it
works on every path and with the reference's own solvers; on the HIP fast path `cg` recognises it and runs the
preconditioned iteration device-resident (`hipk_pcg_solve`: the scaling is fused into the update and direction
kernels, 16 n extra bytes per iteration instead of a separate pass) -- SURVEY 8f-3.
"""
import torch

__all__ = ["JacobiPreconditioner", "BlockJacobiPreconditioner"]


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


def _diagonal_blocks(A: torch.Tensor, bs: int) -> torch.Tensor:
    """[ceil(n / bs), bs, bs]: the diagonal blocks of A; a ragged last block is completed with identity."""
    n = A.shape[0]
    nb = (n + bs - 1) // bs
    if A.layout == torch.strided:
        rows, cols = torch.nonzero(A, as_tuple=True)
        vals = A[rows, cols]
    else:
        coo = A.to_sparse_coo().coalesce() if A.layout != torch.sparse_coo else A.coalesce()
        rows, cols, vals = coo.indices()[0], coo.indices()[1], coo.values()
    on = torch.div(rows, bs, rounding_mode="floor") == torch.div(cols, bs, rounding_mode="floor")
    r, c, v = rows[on], cols[on], vals[on]
    blocks = torch.zeros(nb * bs * bs, dtype=vals.dtype, device=vals.device)
    blocks.index_add_(0, (torch.div(r, bs, rounding_mode="floor") * bs + r % bs) * bs + c % bs, v)
    blocks = blocks.view(nb, bs, bs)
    pad = nb * bs - n
    if pad:
        idx = torch.arange(bs - pad, bs, device=vals.device)
        blocks[nb - 1, idx, idx] = 1.0
    return blocks


class BlockJacobiPreconditioner:
    """M(v) = blockdiag(A)^-1 v with `block_size` x `block_size` diagonal blocks (1 <= block_size <= 32), inverted once.

    A callable for the reference's `M` hook (TSL:849, 908, 922, 351).  On device vectors the apply is a hand-written
    kernel (`hipk_block_jacobi_apply`) that cg / bicgstab / gmres run between their fused kernels on the solver's stream
    (no synchronisation); on CPU tensors it is a batched matmul.  For unknowns ordered with several degrees of freedom per
    node (systems of PDEs) or for strongly anisotropic stencils, where the point-Jacobi diagonal is a poor approximation."""

    def __init__(self, A: torch.Tensor, block_size: int = 4):
        if not (isinstance(A, torch.Tensor) and A.ndim == 2 and A.shape[0] == A.shape[1]):
            raise ValueError("BlockJacobiPreconditioner needs a square matrix tensor")
        if not 1 <= int(block_size) <= 32:
            raise ValueError("block_size must be in [1, 32]")
        self.block_size = int(block_size)
        self.shape = tuple(A.shape)
        blocks = _diagonal_blocks(A.detach(), self.block_size)
        try:
            self.binv = torch.linalg.inv(blocks).contiguous()
        except RuntimeError as e:
            raise ValueError(f"BlockJacobiPreconditioner: a diagonal block is singular ({e})") from None

    def __call__(self, v):
        n, bs = self.shape[0], self.block_size
        if v.shape != (n,):
            raise ValueError(f"BlockJacobiPreconditioner for {n} unknowns applied to a vector of shape {tuple(v.shape)}")
        binv = self.binv if self.binv.dtype == v.dtype else self.binv.to(v.dtype)
        if v.is_cuda and v.dtype in (torch.float64, torch.float32):
            from .. import _hipk
            if binv is not self.binv:
                self.binv = binv          # keep the converted copy: the solver calls with one dtype
            return _hipk.block_jacobi_apply(binv, bs, v.contiguous())
        nb = binv.shape[0]
        vp = torch.zeros(nb * bs, dtype=v.dtype, device=v.device)
        vp[:n] = v
        return torch.bmm(binv.to(v.device), vp.view(nb, bs, 1)).view(-1)[:n]
