"""Minimal PyTree support for Module A's API surface.

Mirrors the interface of the reference's module_a/torch_tree_util.py (tree_flatten :69,
tree_unflatten :130, tree_map :198, tree_reduce :229, Partial :268): containers are
dict (keys visited in sorted order), list, tuple and NamedTuple; anything else is a leaf.
The solvers only use it at the boundary: a PyTree `b` is ravelled once into one flat
vector, so the hot path never walks trees.
"""
from __future__ import annotations

import functools
from typing import Any, Callable, List, Tuple

import torch


class PyTreeDef:
    """Structure of a PyTree: kind + metadata + child defs (a leaf has kind None)."""

    __slots__ = ("kind", "meta", "children", "num_leaves")

    def __init__(self, kind, meta, children):
        self.kind = kind
        self.meta = meta
        self.children = children
        self.num_leaves = 1 if kind is None else sum(c.num_leaves for c in children)

    def unflatten(self, leaves):
        return tree_unflatten(self, leaves)

    def __eq__(self, other):
        return (isinstance(other, PyTreeDef) and self.kind == other.kind and self.meta == other.meta
                and self.children == other.children)

    def __repr__(self):
        if self.kind is None:
            return "*"
        return f"PyTreeDef({self.kind}, {self.meta}, {self.children})"


def _children(node) -> Tuple[Any, Any, List[Any]]:
    if isinstance(node, dict):
        keys = sorted(node.keys())
        return "dict", tuple(keys), [node[k] for k in keys]
    if isinstance(node, tuple):
        if hasattr(node, "_fields"):
            return "namedtuple", type(node), list(node)
        return "tuple", None, list(node)
    if isinstance(node, list):
        return "list", None, list(node)
    return None, None, []


def tree_flatten(tree: Any) -> Tuple[List[Any], PyTreeDef]:
    leaves: List[Any] = []

    def walk(node) -> PyTreeDef:
        kind, meta, kids = _children(node)
        if kind is None:
            leaves.append(node)
            return PyTreeDef(None, None, [])
        return PyTreeDef(kind, meta, [walk(k) for k in kids])

    treedef = walk(tree)
    return leaves, treedef


def tree_unflatten(treedef: PyTreeDef, leaves) -> Any:
    it = iter(leaves)

    def build(d: PyTreeDef):
        if d.kind is None:
            return next(it)
        kids = [build(c) for c in d.children]
        if d.kind == "dict":
            return dict(zip(d.meta, kids))
        if d.kind == "namedtuple":
            return d.meta(*kids)
        if d.kind == "tuple":
            return tuple(kids)
        return kids

    out = build(treedef)
    return out


def tree_leaves(tree: Any) -> List[Any]:
    return tree_flatten(tree)[0]


def tree_structure(tree: Any) -> PyTreeDef:
    return tree_flatten(tree)[1]


def tree_map(f: Callable, tree: Any, *rest: Any) -> Any:
    leaves, treedef = tree_flatten(tree)
    others = [tree_leaves(r) for r in rest]
    for o in others:
        if len(o) != len(leaves):
            raise ValueError("tree_map: trees must have the same structure")
    return tree_unflatten(treedef, [f(*xs) for xs in zip(leaves, *others)])


def tree_reduce(function: Callable, tree: Any, initializer: Any = None) -> Any:
    leaves = tree_leaves(tree)
    if initializer is None:
        return functools.reduce(function, leaves)
    return functools.reduce(function, leaves, initializer)


class Partial:
    """functools.partial look-alike (jax.tree_util.Partial in the reference)."""

    def __init__(self, func, *args, **kwargs):
        self.func = func
        self.args = args
        self.kwargs = kwargs

    def __call__(self, *more_args, **more_kwargs):
        return self.func(*(self.args + more_args), **{**self.kwargs, **more_kwargs})


# ---- ravel helpers used by the solvers (no reference counterpart: the reference
# re-flattens in every vector operation, TSL:185-203)
def tree_ravel(tree: Any) -> Tuple[torch.Tensor, Callable[[torch.Tensor], Any]]:
    """Concatenate all leaves into one 1-D tensor; return it with the inverse map."""
    leaves, treedef = tree_flatten(tree)
    shapes = [leaf.shape for leaf in leaves]
    sizes = [leaf.numel() for leaf in leaves]
    if len(leaves) == 1:
        flat = leaves[0].reshape(-1)
    else:
        flat = torch.cat([leaf.reshape(-1) for leaf in leaves])

    def unravel(v: torch.Tensor) -> Any:
        if len(shapes) == 1:
            return tree_unflatten(treedef, [v.reshape(shapes[0])])
        out, start = [], 0
        for shp, sz in zip(shapes, sizes):
            out.append(v[start:start + sz].reshape(shp))
            start += sz
        return tree_unflatten(treedef, out)

    return flat, unravel
