"""The benchmark / report harness (reference: tests/benchmark.py:149-273, report table README.md:628-634)."""
import os
import re

import pytest
import torch


def test_harness_runs_on_cpu_and_writes_the_readme_table_shape(tmp_path):
    from pytorch_sparse_solver.tests.benchmark import BenchmarkConfig, SparseSolverBenchmark, main
    cfg = BenchmarkConfig(matrix_sizes=[36, 100], matrix_types=["poisson2d", "tridiagonal", "convdiff_csr"], num_runs=1,
                          warmup_runs=0, device="cpu", tol=1e-8, maxiter=500)
    b = SparseSolverBenchmark(cfg)
    res = b.run_all_benchmarks()
    assert len(res) == 2 * 3 * 3
    ok = [r for r in res if not r.error_message]
    assert len(ok) == len(res)
    assert all(r.converged for r in ok if r.matrix_type != "convdiff_csr" or r.method != "cg")   # cg on a nonsymmetric matrix may fail
    table = b.markdown_tables()
    assert "| Matrix Size | Module A (CG) | Module A (BiCGStab) | Module A (GMRES) | CG it/s | SpMV GB/s (CG, effective) |" in table
    assert re.search(r"\| 100x100 \| [0-9.]+ ms \| [0-9.]+ ms \| [0-9.]+ ms \|", table)
    path = b.generate_markdown_report(str(tmp_path))
    assert os.path.exists(path) and any(f.endswith(".csv") for f in os.listdir(tmp_path))
    assert main(["--quick", "--device", "cpu", "--output-dir", str(tmp_path / "cli"), "--methods", "cg"]) == 0


@pytest.mark.gpu
def test_harness_on_gpu_sparse_inputs(tmp_path):
    from pytorch_sparse_solver.tests.benchmark import BenchmarkConfig, SparseSolverBenchmark
    cfg = BenchmarkConfig(matrix_sizes=[10_000, 250_000], matrix_types=["poisson2d_csr", "ldc_csr"], num_runs=1, warmup_runs=1,
                          device="cuda:0", tol=1e-6, maxiter=5000)
    b = SparseSolverBenchmark(cfg)
    res = b.run_all_benchmarks()
    cg = [r for r in res if r.method == "cg" and r.matrix_type == "poisson2d_csr"]
    assert all(r.converged and r.iters_per_s and r.spmv_gbps for r in cg)
    assert "N=250,000 (500x500 grid)" in b.markdown_tables()
