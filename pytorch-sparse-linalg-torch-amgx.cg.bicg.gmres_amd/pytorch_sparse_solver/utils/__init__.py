"""Utilities: backend availability probes and matrix builders/converters."""
from .availability import (
    check_module_a_available,
    check_module_b_available,
    check_module_c_available,
    get_available_backends,
)
from .matrix_utils import (
    dense_to_sparse_csr,
    sparse_coo_to_csr,
    ensure_sparse_format,
)

__all__ = [
    'check_module_a_available', 'check_module_b_available', 'check_module_c_available',
    'get_available_backends', 'dense_to_sparse_csr', 'sparse_coo_to_csr', 'ensure_sparse_format',
]
