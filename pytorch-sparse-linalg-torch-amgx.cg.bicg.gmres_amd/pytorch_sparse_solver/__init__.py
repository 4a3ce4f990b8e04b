"""pytorch_sparse_solver -- MI355X-native build of the Module-A iterative-solver hot path.

Import surface of the reference package (`__init__.py:46-113`): SparseSolver, solve,
cg/bicgstab/gmres shortcuts, availability probes and matrix utilities.
"""
__version__ = '1.0.0'

from .solver import (
    SparseSolver, SolverResult, SolverMethod, SolverBackend,
    solve, cg, bicgstab, gmres, amg, direct_solve,
)
from .utils.availability import (
    check_module_a_available, check_module_b_available, check_module_c_available,
    get_available_backends, print_availability_report,
)
from .utils.matrix_utils import (
    dense_to_sparse_csr, sparse_coo_to_csr, ensure_sparse_format,
    create_tridiagonal_sparse_coo, create_poisson_2d_sparse_coo,
    compute_residual, compute_relative_residual,
)

__author__ = 'pytorch_sparse_solver for MI355X contributors'
__license__ = 'Apache-2.0'

__all__ = [
    '__version__', '__author__', '__license__',
    'SparseSolver', 'SolverResult', 'SolverMethod', 'SolverBackend',
    'solve', 'cg', 'bicgstab', 'gmres', 'amg', 'direct_solve',
    'check_module_a_available', 'check_module_b_available', 'check_module_c_available',
    'get_available_backends', 'print_availability_report',
    'dense_to_sparse_csr', 'sparse_coo_to_csr', 'ensure_sparse_format',
    'create_tridiagonal_sparse_coo', 'create_poisson_2d_sparse_coo',
    'compute_residual', 'compute_relative_residual',
]


def __getattr__(name):
    # not part of the reference's import surface: the row-partitioned operand (one process per GPU), imported on first use
    if name == 'RowBlockCSR':
        from .distributed import RowBlockCSR
        return RowBlockCSR
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
