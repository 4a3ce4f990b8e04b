"""Benchmark harness of the package (`python -m pytorch_sparse_solver.tests.benchmark`), as in the reference layout."""
