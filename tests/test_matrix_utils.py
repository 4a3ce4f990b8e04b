"""utils/matrix_utils.py: the vectorised builders against arrays captured from the reference's own (loop) builders
(tests/golden/*.npz: poisson_*, ldc_*, convdiff_* store the reference-built CSR), plus the format helpers."""
import numpy as np
import pytest
import torch

from conftest import load_case
from pytorch_sparse_solver.utils import matrix_utils as mu


def _same(A, d):
    return (np.array_equal(A.crow_indices().numpy(), d["crow"]) and np.array_equal(A.col_indices().numpy(), d["col"])
            and np.array_equal(A.values().numpy(), d["val"]))


@pytest.mark.parametrize("nx", [8, 16, 32, 64])
def test_poisson_builder_bit_identical_to_reference(nx):
    d = load_case(f"poisson_nx{nx}")                      # built by the reference's create_poisson_2d_sparse_coo
    assert _same(mu.create_poisson_2d_sparse_coo(nx, nx).to_sparse_csr(), d)
    assert _same(mu.create_poisson_2d_csr(nx, nx), d)
    A = mu.create_poisson_2d_csr(nx, nx)
    assert A.crow_indices().dtype == torch.int64 and A.values().dtype == torch.float64


def test_poisson_builder_ragged_grid():
    assert _same(mu.create_poisson_2d_sparse_coo(17, 13).to_sparse_csr(), load_case("poisson_17x13"))


@pytest.mark.parametrize("nx", [8, 16, 32])
def test_ldc_pressure_builder_bit_identical_to_reference(nx):
    d = load_case(f"ldc_nx{nx}_step0")                    # built by the reference's BaseLDCSolver._setup_pressure_matrix
    A = mu.create_ldc_pressure_csr(nx)
    assert _same(A, d)
    assert torch.allclose(A.to_dense().sum(dim=1), torch.zeros(nx * nx, dtype=torch.float64), atol=1e-9)   # singular: row sums 0
    assert torch.equal(A.to_dense(), A.to_dense().T)


@pytest.mark.parametrize("nx", [16, 32, 64])
def test_convdiff_builder_matches_fixture(nx):
    assert _same(mu.create_convdiff_2d_csr(nx, nx), load_case(f"convdiff_nx{nx}"))


def test_row_block_builder_matches_global():
    crow, col, val = mu.stencil5_csr_components(12, 9, 4.0, -1.0, -1.0, -1.0, -1.0)
    c2, col2, val2 = mu.stencil5_csr_components(12, 9, 4.0, -1.0, -1.0, -1.0, -1.0, row_begin=30, row_end=77)
    j0, j1 = int(crow[30]), int(crow[77])
    assert torch.equal(c2, crow[30:78] - j0) and torch.equal(col2, col[j0:j1]) and torch.equal(val2, val[j0:j1])


def test_format_helpers():
    A = mu.create_tridiagonal_sparse_coo(6)
    D = A.to_dense()
    assert D[0, 0] == 2 and D[0, 1] == -1 and D[5, 4] == -1 and D[0, 5] == 0
    for fmt, layout in (("csr", torch.sparse_csr), ("coo", torch.sparse_coo), ("csc", torch.sparse_csc)):
        assert mu.ensure_sparse_format(D, fmt).layout == layout
        assert torch.equal(mu.ensure_sparse_format(D, fmt).to_dense(), D)
    with pytest.raises(ValueError, match="Unknown format"):
        mu.ensure_sparse_format(D, "ell")
    assert mu.dense_to_sparse_csr(D).layout == torch.sparse_csr
    with pytest.raises(ValueError, match="2D"):
        mu.dense_to_sparse_csr(torch.zeros(3))
    assert mu.sparse_coo_to_csr(A).layout == torch.sparse_csr
    with pytest.raises(ValueError, match="sparse"):
        mu.sparse_coo_to_csr(D)
    v, c, r = mu.get_csr_components(D)
    B = mu.create_sparse_csr_from_components(v, c, r, (6, 6))
    assert torch.equal(B.to_dense(), D)
    x = torch.arange(6, dtype=torch.float64)
    b = D @ x
    assert mu.compute_relative_residual(D, x, b) == 0.0
    assert mu.compute_relative_residual(mu.dense_to_sparse_csr(D), x, b) < 1e-15
    assert mu.compute_relative_residual(A, x, b) < 1e-15
    assert torch.equal(mu.compute_residual(lambda v_: D @ v_, x, b), torch.zeros(6, dtype=torch.float64))
