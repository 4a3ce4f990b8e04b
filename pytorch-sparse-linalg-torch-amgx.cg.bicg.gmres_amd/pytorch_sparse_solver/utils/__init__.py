"""Utilities of the hot path's callers: backend probes (availability.py) and the matrix
builders / layout converters (matrix_utils.py).  Names re-exported here are the ones the
reference's `utils` package exposes, plus the stencil builders the benchmark and tests use."""
from . import availability as _av
from . import matrix_utils as _mu

_PROBES = ("check_module_a_available", "check_module_b_available", "check_module_c_available",
           "get_available_backends")
_CONVERTERS = ("dense_to_sparse_csr", "sparse_coo_to_csr", "ensure_sparse_format")
_BUILDERS = ("create_poisson_2d_csr", "create_poisson_2d_sparse_coo", "create_convdiff_2d_csr",
             "create_ldc_pressure_csr", "create_variable_diffusion_2d_csr", "stencil5_csr_components")

for _name in _PROBES:
    globals()[_name] = getattr(_av, _name)
for _name in _CONVERTERS + _BUILDERS:
    globals()[_name] = getattr(_mu, _name)
del _name

__all__ = list(_PROBES + _CONVERTERS + _BUILDERS)
