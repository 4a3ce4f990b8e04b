"""Module A: JAX-style iterative solvers (cg, bicgstab, gmres), MI355X-native.

Same exports as the reference's `module_a/__init__.py:47-63`; CUDA/ROCm tensor inputs run
on hand-written gfx950 kernels (libhipk.so), everything else on the generic torch path.
`get_last_stats()` (iteration counts the reference never returns) and `JacobiPreconditioner` (a callable for the
reference's `M` hook that the fast path runs device-resident) are the additions.
"""
from .torch_sparse_linalg import (
    cg, bicgstab, gmres,
    cg_differentiable, bicgstab_differentiable, gmres_differentiable,
    LinearSolveFunction, ImplicitAdjointFunction, get_last_stats,
)
from .torch_tree_util import tree_leaves, tree_map, tree_flatten, tree_unflatten, Partial
from .preconditioners import BlockJacobiPreconditioner, JacobiPreconditioner

__all__ = [
    'cg', 'bicgstab', 'gmres',
    'cg_differentiable', 'bicgstab_differentiable', 'gmres_differentiable',
    'LinearSolveFunction',
    'tree_leaves', 'tree_map', 'tree_flatten', 'tree_unflatten', 'Partial',
    'get_last_stats', 'JacobiPreconditioner', 'BlockJacobiPreconditioner',
]

__version__ = '1.0.0'
