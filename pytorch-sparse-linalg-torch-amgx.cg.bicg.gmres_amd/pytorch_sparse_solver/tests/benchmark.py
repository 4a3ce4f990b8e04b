"""Benchmark / report harness of Module A on this build.

Counterpart of the reference's `tests/benchmark.py:149-273` + the report its `run.py:174-278` writes: the same protocol
(matrix recipe -> `SparseSolver.solve(A, b, method, backend='module_a', tol, maxiter)`, `warmup_runs` untimed solves, mean of
`num_runs` timed ones with a device synchronise around each, relative residual and convergence verdict per cell) and the same
report shape -- the "Matrix Size | Module A (CG) | Module A (GMRES) ..." table of README.md:628-634 -- extended with what
the hot path is judged on here: SPARSE inputs at production sizes, iterations/s and SpMV GB/s columns.

    python -m pytorch_sparse_solver.tests.benchmark --quick                      # the reference's quick sizes, dense inputs
    python -m pytorch_sparse_solver.tests.benchmark --sparse --sizes 1000000,4000000
    python -m pytorch_sparse_solver.tests.benchmark --sizes 100,500 --runs 5 --output-dir Logger

Matrix types: `tridiagonal`, `poisson2d`, `dense_spd` (dense tensors, what the reference benchmarks: they are converted to
CSR once by the handle cache) and `poisson2d_csr`, `convdiff_csr`, `ldc_csr` (built directly in CSR; the BASELINE configs).
Writes `benchmark_report_<timestamp>.md` and `benchmark_results_<timestamp>.csv` into the output directory.
"""
from __future__ import annotations

import argparse
import csv
import math
import os
import time
from dataclasses import asdict, dataclass, field
from datetime import datetime
from typing import List, Optional

import torch

DENSE_TYPES = ("tridiagonal", "poisson2d", "dense_spd")
SPARSE_TYPES = ("poisson2d_csr", "convdiff_csr", "ldc_csr")


@dataclass
class BenchmarkResult:
    backend: str
    method: str
    matrix_size: int
    matrix_type: str
    solve_time: float            # seconds, mean of the timed runs
    residual: float
    converged: bool
    iterations: Optional[int] = None          # solver iterations (GMRES: restart cycles) from get_last_stats()
    matvecs: Optional[int] = None
    iters_per_s: Optional[float] = None
    spmv_gbps: Optional[float] = None         # CSR-formula bytes x operator applications / solve time (effective)
    nnz: Optional[int] = None
    error_message: Optional[str] = None


@dataclass
class BenchmarkConfig:
    matrix_sizes: List[int] = field(default_factory=lambda: [100, 500, 1000])
    methods: List[str] = field(default_factory=lambda: ["cg", "bicgstab", "gmres"])
    matrix_types: List[str] = field(default_factory=lambda: list(DENSE_TYPES))
    num_runs: int = 3
    warmup_runs: int = 1
    device: str = "cuda" if torch.cuda.is_available() else "cpu"
    dtype: torch.dtype = torch.float64
    tol: float = 1e-8
    maxiter: int = 1000
    restart: int = 30


def create_matrix(n: int, matrix_type: str, device, dtype):
    """(A, b, effective n).  RHS = A x_true with x_true ~ randn, as the reference's harness does."""
    from ..utils.matrix_utils import (create_convdiff_2d_csr, create_ldc_pressure_csr, create_poisson_2d_csr,
                                      create_poisson_2d_sparse_coo)
    g = max(2, int(math.isqrt(n)))
    if matrix_type == "tridiagonal":
        A = (2.0 * torch.eye(n, dtype=dtype) - torch.diag(torch.ones(n - 1, dtype=dtype), 1)
             - torch.diag(torch.ones(n - 1, dtype=dtype), -1)).to(device)
    elif matrix_type == "poisson2d":
        A = create_poisson_2d_sparse_coo(g, g, dtype=dtype).to_dense().to(device)
    elif matrix_type == "dense_spd":
        G = torch.randn(n, n, dtype=dtype, generator=torch.Generator().manual_seed(n))
        A = (G @ G.T + n * torch.eye(n, dtype=dtype)).to(device)
    elif matrix_type == "poisson2d_csr":
        A = create_poisson_2d_csr(g, g, device=device, dtype=dtype)
    elif matrix_type == "convdiff_csr":
        A = create_convdiff_2d_csr(g, g, device=device, dtype=dtype)
    elif matrix_type == "ldc_csr":
        A = create_ldc_pressure_csr(g, device=device, dtype=dtype)
    else:
        raise ValueError(f"Unknown matrix type: {matrix_type}")
    m = A.shape[0]
    x_true = torch.randn(m, dtype=dtype, generator=torch.Generator().manual_seed(m + 1)).to(device)
    if matrix_type == "ldc_csr":
        x_true -= x_true.mean()          # the Neumann pressure matrix is singular: consistent right-hand side
    b = A @ x_true
    return A, b, m


def _nnz(A) -> int:
    if A.layout == torch.strided:
        return int((A != 0).sum().item())
    return int(A.values().numel() if A.layout == torch.sparse_csr else A._nnz())


class SparseSolverBenchmark:
    def __init__(self, config: BenchmarkConfig):
        self.config = config
        self.results: List[BenchmarkResult] = []
        from .. import SparseSolver
        self.solver = SparseSolver(verbose=False)

    def _sync(self):
        if str(self.config.device).startswith("cuda"):
            torch.cuda.synchronize()

    def run_single_benchmark(self, method: str, matrix_size: int, matrix_type: str) -> BenchmarkResult:
        from ..module_a import get_last_stats
        cfg = self.config
        try:
            A, b, n = create_matrix(matrix_size, matrix_type, cfg.device, cfg.dtype)
        except Exception as e:  # noqa: BLE001
            return BenchmarkResult("module_a", method, matrix_size, matrix_type, 0.0, float("inf"), False,
                                   error_message=f"Matrix creation failed: {e}")
        kw = {"restart": cfg.restart} if method == "gmres" else {}
        try:
            for _ in range(cfg.warmup_runs):
                self.solver.solve(A, b, method=method, backend="module_a", tol=cfg.tol, maxiter=cfg.maxiter, **kw)
            times = []
            for _ in range(cfg.num_runs):
                self._sync()
                t0 = time.perf_counter()
                x, res = self.solver.solve(A, b, method=method, backend="module_a", tol=cfg.tol, maxiter=cfg.maxiter, **kw)
                self._sync()
                times.append(time.perf_counter() - t0)
        except Exception as e:  # noqa: BLE001
            return BenchmarkResult("module_a", method, n, matrix_type, 0.0, float("inf"), False, error_message=str(e))
        st = get_last_stats()
        dt = sum(times) / len(times)
        nnz = _nnz(A)
        sv = 8 if cfg.dtype == torch.float64 else 4
        spmv_bytes = nnz * (sv + 4) + (n + 1) * 4 + 2 * n * sv           # SURVEY 8d formula
        its, mv = getattr(st, "iterations", None), getattr(st, "matvecs", None)
        return BenchmarkResult("module_a", method, n, matrix_type, dt, float(res.residual), bool(res.converged),
                               iterations=its, matvecs=mv, iters_per_s=(its / dt) if its else None,
                               spmv_gbps=(spmv_bytes * mv / dt / 1e9) if mv else None, nnz=nnz)

    def run_all_benchmarks(self) -> List[BenchmarkResult]:
        cfg = self.config
        print("=" * 80 + f"\nModule A benchmark on {cfg.device}: sizes {cfg.matrix_sizes}, methods {cfg.methods}, "
              f"matrix types {cfg.matrix_types}, {cfg.num_runs} runs + {cfg.warmup_runs} warm-up\n" + "=" * 80)
        for mt in cfg.matrix_types:
            for size in cfg.matrix_sizes:
                for method in cfg.methods:
                    r = self.run_single_benchmark(method, size, mt)
                    self.results.append(r)
                    state = f"SKIP ({r.error_message[:40]})" if r.error_message else ("OK" if r.converged else "NOT CONVERGED")
                    print(f"  {mt:14s} n={r.matrix_size:<9d} {method:9s} {state:14s} {r.solve_time * 1e3:10.3f} ms  "
                          f"residual {r.residual:.2e}", flush=True)
        return self.results

    def export_csv(self, filename: str) -> None:
        with open(filename, "w", newline="") as f:
            w = csv.writer(f)
            cols = list(asdict(self.results[0]).keys()) if self.results else []
            w.writerow(cols)
            for r in self.results:
                w.writerow([asdict(r)[c] for c in cols])

    def markdown_tables(self) -> str:
        """One table per matrix type in the shape of README.md:628-634, plus the throughput columns."""
        out = []
        heads = {"cg": "Module A (CG)", "bicgstab": "Module A (BiCGStab)", "gmres": "Module A (GMRES)"}
        for mt in self.config.matrix_types:
            rows = [r for r in self.results if r.matrix_type == mt]
            if not rows:
                continue
            methods = [m for m in self.config.methods if any(r.method == m for r in rows)]
            out.append(f"### {mt}\n")
            out.append("| Matrix Size | " + " | ".join(heads.get(m, m) for m in methods) + " | CG it/s | SpMV GB/s (CG, effective) |")
            out.append("|" + "---|" * (len(methods) + 3))
            for size in sorted({r.matrix_size for r in rows}):
                cells = []
                for m in methods:
                    r = next((r for r in rows if r.matrix_size == size and r.method == m), None)
                    if r is None or r.error_message:
                        cells.append("n/a")
                    else:
                        cells.append(f"{r.solve_time * 1e3:.1f} ms" + ("" if r.converged else " (nc)"))
                cgr = next((r for r in rows if r.matrix_size == size and r.method == "cg" and not r.error_message), None)
                its = f"{cgr.iters_per_s:,.0f}" if cgr and cgr.iters_per_s else "n/a"
                gb = f"{cgr.spmv_gbps:,.1f}" if cgr and cgr.spmv_gbps else "n/a"
                side = int(math.isqrt(size))
                label = f"{size}x{size}" if size < 10_000 else f"N={size:,} ({side}x{side} grid)"
                out.append(f"| {label} | " + " | ".join(cells) + f" | {its} | {gb} |")
            out.append("")
        return "\n".join(out)

    def generate_markdown_report(self, output_dir: str) -> str:
        os.makedirs(output_dir, exist_ok=True)
        stamp = datetime.now().strftime("%Y-%m-%d_%H-%M-%S")
        path = os.path.join(output_dir, f"benchmark_report_{stamp}.md")
        cfg = self.config
        dev = torch.cuda.get_device_name(0) if str(cfg.device).startswith("cuda") else "cpu"
        ok = [r for r in self.results if not r.error_message]
        with open(path, "w") as f:
            f.write(f"# Module A benchmark report\n\n*{stamp}* -- device **{dev}**, torch {torch.__version__}, dtype {cfg.dtype}, "
                    f"tol {cfg.tol:g}, maxiter {cfg.maxiter}, GMRES restart {cfg.restart}, {cfg.num_runs} timed runs after "
                    f"{cfg.warmup_runs} warm-up; protocol of the reference's `tests/benchmark.py`.  `(nc)` = not converged within "
                    f"maxiter.  SpMV GB/s = CSR-formula bytes x operator applications / solve time (an effective figure).\n\n")
            f.write(self.markdown_tables())
            f.write(f"\n{sum(r.converged for r in ok)} of {len(ok)} cells converged; {len(self.results) - len(ok)} skipped.\n")
        self.export_csv(os.path.join(output_dir, f"benchmark_results_{stamp}.csv"))
        return path


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description="Module A benchmark (reference protocol, README table shape)")
    ap.add_argument("--quick", action="store_true", help="the reference's quick sizes 100,200,500 with 2 runs")
    ap.add_argument("--sizes", type=str, default=None, help="comma-separated matrix sizes (unknowns)")
    ap.add_argument("--runs", type=int, default=None)
    ap.add_argument("--sparse", action="store_true", help="CSR-built matrices (poisson2d_csr, convdiff_csr, ldc_csr)")
    ap.add_argument("--types", type=str, default=None, help="comma-separated matrix types")
    ap.add_argument("--methods", type=str, default="cg,bicgstab,gmres")
    ap.add_argument("--tol", type=float, default=1e-8)
    ap.add_argument("--maxiter", type=int, default=1000)
    ap.add_argument("--device", type=str, default=None)
    ap.add_argument("--output-dir", type=str, default="Logger")
    a = ap.parse_args(argv)
    cfg = BenchmarkConfig(tol=a.tol, maxiter=a.maxiter, methods=a.methods.split(","))
    if a.quick:
        cfg.matrix_sizes, cfg.num_runs, cfg.matrix_types = [100, 200, 500], 2, ["poisson2d"]
    if a.sparse:
        cfg.matrix_types = list(SPARSE_TYPES)
    if a.types:
        cfg.matrix_types = a.types.split(",")
    if a.sizes:
        cfg.matrix_sizes = [int(s) for s in a.sizes.split(",")]
    if a.runs:
        cfg.num_runs = a.runs
    if a.device:
        cfg.device = a.device
    bench = SparseSolverBenchmark(cfg)
    bench.run_all_benchmarks()
    path = bench.generate_markdown_report(a.output_dir)
    print(f"\nreport: {path}\n")
    print(bench.markdown_tables())
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
