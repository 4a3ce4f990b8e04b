"""Matrix builders and format helpers (reference: utils/matrix_utils.py).

Same public functions as the reference; the stencil builders are vectorised (the
reference's Python double loop, matrix_utils.py:193-257, needs minutes at N = 4M) and
produce bit-identical COO/CSR tensors (checked in tests/test_matrix_utils.py against
arrays captured from the reference).  New here: direct CSR builders for the benchmark
matrices (5-point Poisson, convection-diffusion, LDC pressure), including row blocks for
the row-partitioned multi-GPU solver.
"""
from typing import Optional, Tuple, Union

import torch


def dense_to_sparse_csr(A: torch.Tensor, device: Optional[str] = None) -> torch.Tensor:
    if A.ndim != 2:
        raise ValueError(f"Expected 2D tensor, got {A.ndim}D")
    coo = A.to_sparse_coo()
    if device is not None and torch.device(device) != A.device:
        coo = coo.to(device)
    return coo.to_sparse_csr()


def sparse_coo_to_csr(sparse_coo: torch.Tensor) -> torch.Tensor:
    if not sparse_coo.is_sparse:
        raise ValueError("Input must be a sparse tensor")
    return sparse_coo.coalesce().to_sparse_csr()


def ensure_sparse_format(A: torch.Tensor, format: str = 'csr') -> torch.Tensor:
    if A.layout == torch.sparse_csr:
        if format == 'csr':
            return A
        A = A.to_sparse_coo()
    elif not A.is_sparse:
        A = A.to_sparse_coo()
    if format == 'csr':
        return A.coalesce().to_sparse_csr()
    if format == 'coo':
        return A.coalesce()
    if format == 'csc':
        return A.coalesce().to_sparse_csc()
    raise ValueError(f"Unknown format: {format}. Use 'csr', 'coo', or 'csc'")


def get_csr_components(A: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """(values, col_indices, row_ptr) of A, converting to CSR if necessary."""
    if A.layout != torch.sparse_csr:
        A = ensure_sparse_format(A, 'csr')
    return A.values(), A.col_indices(), A.crow_indices()


def create_sparse_csr_from_components(values, col_indices, row_ptr, shape, device=None, dtype=None) -> torch.Tensor:
    device = values.device if device is None else device
    dtype = values.dtype if dtype is None else dtype
    return torch.sparse_csr_tensor(crow_indices=row_ptr.to(device), col_indices=col_indices.to(device),
                                   values=values.to(device=device, dtype=dtype), size=shape)


def create_tridiagonal_sparse_coo(n: int, diag_val: float = 2.0, off_diag_val: float = -1.0,
                                  device: str = 'cpu', dtype: torch.dtype = torch.float64) -> torch.Tensor:
    i = torch.arange(n, device=device)
    rows, cols = [i], [i]
    vals = [torch.full((n,), diag_val, device=device, dtype=dtype)]
    if n > 1:
        j = i[:-1]
        rows += [j, j + 1]
        cols += [j + 1, j]
        vals += [torch.full((n - 1,), off_diag_val, device=device, dtype=dtype)] * 2
    idx = torch.stack([torch.cat(rows), torch.cat(cols)])
    return torch.sparse_coo_tensor(idx, torch.cat(vals), (n, n), device=device, dtype=dtype).coalesce()


# ----------------------------------------------------------------------------- 5-point stencils
def stencil5_csr_components(nx: int, ny: int, center, west, east, south, north, *, row_begin: int = 0,
                            row_end: Optional[int] = None, device='cpu', dtype=torch.float64,
                            index_dtype=torch.int64):
    """CSR arrays of rows [row_begin, row_end) of a 5-point operator on an nx x ny grid.

    Row k = i*ny + j (i in [0,nx), j in [0,ny)) couples to k-ny (`west`, i-1), k-1 (`south`,
    j-1), k (`center`), k+1 (`north`, j+1), k+ny (`east`, i+1) when the neighbour exists
    (Dirichlet truncation), columns ascending -- the ordering of the reference's
    `create_poisson_2d_sparse_coo` (matrix_utils.py:193-257) after coalescing.
    Coefficients are python floats or callables (i, j) -> tensor.
    Column indices are GLOBAL.  Returns (crow, col, val).
    """
    n = nx * ny
    row_end = n if row_end is None else row_end
    k = torch.arange(row_begin, row_end, device=device, dtype=torch.int64)
    i, j = torch.div(k, ny, rounding_mode='floor'), k % ny

    def coef(c):
        if callable(c):
            return c(i, j).to(dtype)
        return torch.full((k.numel(),), float(c), device=device, dtype=dtype)

    has = [i > 0, j > 0, torch.ones_like(i, dtype=torch.bool), j < ny - 1, i < nx - 1]
    off = [-ny, -1, 0, 1, ny]
    cf = [coef(west), coef(south), coef(center), coef(north), coef(east)]
    mask = torch.stack(has, dim=1)                                  # (rows, 5) in ascending-column order
    cols = torch.stack([k + o for o in off], dim=1)
    vals = torch.stack(cf, dim=1)
    counts = mask.sum(dim=1)
    crow = torch.zeros(k.numel() + 1, device=device, dtype=torch.int64)
    torch.cumsum(counts, dim=0, out=crow[1:])
    return crow.to(index_dtype), cols[mask].to(index_dtype), vals[mask]


def create_poisson_2d_csr(nx: int, ny: int, device='cpu', dtype=torch.float64) -> torch.Tensor:
    """5-point Poisson matrix (diag 4, neighbours -1) directly in CSR."""
    crow, col, val = stencil5_csr_components(nx, ny, 4.0, -1.0, -1.0, -1.0, -1.0, device=device, dtype=dtype)
    return torch.sparse_csr_tensor(crow, col, val, size=(nx * ny, nx * ny))


def create_poisson_2d_sparse_coo(nx: int, ny: int, device: str = 'cpu',
                                 dtype: torch.dtype = torch.float64) -> torch.Tensor:
    """Reference API (matrix_utils.py:193-257): coalesced COO 5-point Poisson matrix."""
    crow, col, val = stencil5_csr_components(nx, ny, 4.0, -1.0, -1.0, -1.0, -1.0, device=device, dtype=dtype)
    n = nx * ny
    rows = torch.repeat_interleave(torch.arange(n, device=device), crow[1:] - crow[:-1])
    idx = torch.stack([rows, col])
    return torch.sparse_coo_tensor(idx, val, (n, n), device=device, dtype=dtype).coalesce()


def create_convdiff_2d_csr(nx: int, ny: int, gamma: float = 0.5, delta: float = 0.25, device='cpu',
                           dtype=torch.float64) -> torch.Tensor:
    """Nonsymmetric convection-diffusion test matrix (BASELINE config 3):
    diag 4, west -1-gamma, east -1+gamma, south -1-delta, north -1+delta."""
    crow, col, val = stencil5_csr_components(nx, ny, 4.0, -1.0 - gamma, -1.0 + gamma, -1.0 - delta, -1.0 + delta,
                                             device=device, dtype=dtype)
    return torch.sparse_csr_tensor(crow, col, val, size=(nx * ny, nx * ny))


def create_ldc_pressure_csr(nx: int, device='cpu', dtype=torch.float64) -> torch.Tensor:
    """Pressure-Poisson matrix of the lid-driven-cavity example
    (FVM_example/LDC_by_torchsp/ldc_solver_common.py:90-135): n = nx^2, row i = iy*nx + ix,
    Neumann 5-point Laplacian scaled by 1/dx^2 (singular, row sums 0)."""
    ny = nx
    dx2 = (1.0 / nx) ** 2
    c = 1.0 / dx2
    # in stencil5 terms the slow index is iy ("i"), the fast index ix ("j")
    def diag(i, j):
        aw = (j > 0).to(torch.float64) * c
        ae = (j < nx - 1).to(torch.float64) * c
        an = (i < ny - 1).to(torch.float64) * c
        as_ = (i > 0).to(torch.float64) * c
        return -(aw + ae + an + as_)
    crow, col, val = stencil5_csr_components(ny, nx, diag, c, c, c, c, device=device, dtype=dtype)
    return torch.sparse_csr_tensor(crow, col, val, size=(nx * nx, nx * nx))


# ----------------------------------------------------------------------------- residual helpers
def _apply(A, x):
    if callable(A):
        return A(x)
    if A.layout == torch.strided:
        return torch.mv(A, x)
    if A.is_sparse:
        return torch.sparse.mm(A, x.unsqueeze(-1)).squeeze(-1)
    return torch.matmul(A, x)


def compute_residual(A: Union[torch.Tensor, callable], x: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    return b - _apply(A, x)


def compute_relative_residual(A: Union[torch.Tensor, callable], x: torch.Tensor, b: torch.Tensor) -> float:
    return (torch.norm(compute_residual(A, x, b)) / torch.norm(b)).item()


def create_variable_diffusion_2d_csr(nx: int, ny: int, contrast: float = 2.0, seed: int = 0, device='cpu',
                                     dtype=torch.float64) -> torch.Tensor:
    """-div(k grad u) on an nx x ny grid (row k = i*ny + j, Dirichlet truncation) with a log-normal cell coefficient
    k = exp(contrast * N(0,1)) and harmonic face averages: SPD, 5-point, strongly varying diagonal -- the test
    problem of the Jacobi-preconditioned CG path (not in the reference; synthetic)."""
    g = torch.Generator().manual_seed(seed)
    k = torch.exp(contrast * torch.randn(nx + 2, ny + 2, generator=g, dtype=torch.float64))
    kc = k[1:-1, 1:-1]

    def harm(a, b):
        return 2.0 * a * b / (a + b)

    faces = ((harm(kc, k[:-2, 1:-1]), (-1, 0)), (harm(kc, k[2:, 1:-1]), (1, 0)),
             (harm(kc, k[1:-1, :-2]), (0, -1)), (harm(kc, k[1:-1, 2:]), (0, 1)))
    idx = torch.arange(nx * ny).reshape(nx, ny)
    rows, cols, vals = [idx.reshape(-1)], [idx.reshape(-1)], [sum(w for w, _ in faces).reshape(-1)]
    for w, (di, dj) in faces:
        i0, i1 = max(0, -di), nx - max(0, di)
        j0, j1 = max(0, -dj), ny - max(0, dj)
        rows.append(idx[i0:i1, j0:j1].reshape(-1))
        cols.append(idx[i0 + di:i1 + di, j0 + dj:j1 + dj].reshape(-1))
        vals.append(-w[i0:i1, j0:j1].reshape(-1))
    A = torch.sparse_coo_tensor(torch.stack([torch.cat(rows), torch.cat(cols)]), torch.cat(vals).to(dtype),
                                (nx * ny, nx * ny))
    return A.coalesce().to_sparse_csr().to(device)
