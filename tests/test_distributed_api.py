"""The row-partitioned solvers BEHIND THE REFERENCE'S CALL SURFACE: every rank calls only
`SparseSolver().solve(A, b, method, backend='module_a')` (solver.py:256-379) or `module_a.cg / bicgstab / gmres`
(TSL:1019, 1091, 641) with a `RowBlockCSR` operand -- its rows of the global matrix -- and its slices of b / x0, and gets the
single-rank solve's bits.  CPU: gloo world 2 / 3 with the CPU ops double (cg).  GPU: ranks share cuda:0, the C-driven loops
(cg, bicgstab, gmres) with host-staged collectives."""
import json
import os
import socket
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(world, kind, nx, ny, tol, maxiter, tmp_path, mode, solver, entry):
    out = str(tmp_path / f"api_{world}_{kind}_{solver}_{entry}.json")
    for _attempt in range(3):   # a port found free can be taken before the store binds it (EADDRINUSE): try another one
        port = _free_port()
        procs = []
        for r in range(world):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port), OMP_NUM_THREADS="1")
            procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_dist_api_worker.py"), kind, str(nx), str(ny), str(tol),
                                           str(maxiter), out, mode, solver, entry], env=env, stdout=subprocess.PIPE,
                                          stderr=subprocess.STDOUT))
        logs = []
        for p in procs:
            try:
                o, _ = p.communicate(timeout=300)
            except subprocess.TimeoutExpired:
                for q in procs:
                    q.kill()
                raise
            logs.append(o.decode(errors="replace"))
        if all(p.returncode == 0 for p in procs) or not any("EADDRINUSE" in lg for lg in logs):
            break
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    with open(out) as f:
        return json.load(f)


def _check(r, entry, method):
    assert r["bitwise_equal"], r
    assert set(r["info"]) == {r["ref_info"]} and set(r["iterations"]) == {r["ref_iterations"]}
    assert set(r["residual_norm"]) == {r["ref_residual_norm"]}
    assert r["second_bitwise_equal"] and set(r["second_info"]) == {r["ref2_info"]}     # cached plan, warm start
    if entry == "solver":
        for rec in r["records"]:       # the reference's result record (solver.py:73-81, 356-379), the same on every rank
            assert rec["backend"] == "module_a" and rec["method"] == method and rec["iterations"] is None
            assert rec["converged"] == (r["ref_info"] == 0)
            assert abs(rec["residual"] - r["ref_residual_norm"] / r["ref_b_norm"]) <= 1e-15 * max(rec["residual"], 1e-300) + 1e-300


@pytest.mark.parametrize("world,kind,nx,ny,entry", [
    (2, "poisson", 96, 64, "solver"), (3, "poisson", 96, 64, "solver"), (2, "random_spd", 80, 77, "module_a"),
    (3, "random_spd", 80, 77, "solver"), (4, "poisson", 96, 64, "module_a"),     # rank 3 owns no rows
    (2, "poisson", 4, 8000, "solver")])
def test_cg_through_the_reference_call_surface_gloo(world, kind, nx, ny, entry, tmp_path):
    r = _run(world, kind, nx, ny, 1e-8, -1, tmp_path, "cpu", "cg", entry)
    _check(r, entry, "cg")
    assert sum(r["n_local"]) == nx * ny


def test_cg_maxiter_cutoff_through_the_call_surface_gloo(tmp_path):
    r = _run(2, "poisson", 96, 64, 1e-12, 9, tmp_path, "cpu", "cg", "solver")
    assert r["bitwise_equal"] and set(r["iterations"]) == {9} and set(r["info"]) == {-1}
    assert all(rec["converged"] is False for rec in r["records"])


def test_row_block_operand_errors():
    """No process group: a one-rank block.  The operand's misuse raises ValueError with the reference's wording where it has one."""
    import torch
    from pytorch_sparse_solver import RowBlockCSR
    from pytorch_sparse_solver.module_a import cg, gmres
    from pytorch_sparse_solver.utils.matrix_utils import create_poisson_2d_csr
    A = create_poisson_2d_csr(8, 8)
    assert RowBlockCSR.row_range(64) == (0, 64)
    Arb = RowBlockCSR.from_global_csr(A)
    b = torch.ones(64, dtype=torch.float64)
    with pytest.raises(ValueError, match="HIP kernels"):
        cg(Arb, b)                                               # CPU tensors without an ops backend
    with pytest.raises(ValueError, match="matching shapes"):
        cg(Arb, b, x0=torch.zeros(63, dtype=torch.float64))
    with pytest.raises(ValueError, match="slice of the right-hand side"):
        cg(Arb, torch.ones(65, dtype=torch.float64))
    with pytest.raises(ValueError, match="preconditioners"):
        cg(Arb, b, M=lambda v: v)
    with pytest.raises(ValueError, match="Unsupported solve_method"):
        gmres(Arb, b, solve_method="nope")
    with pytest.raises(ValueError, match="row pointers"):
        RowBlockCSR(A.crow_indices()[:-1], A.col_indices(), A.values(), 64)


@pytest.mark.gpu
@pytest.mark.parametrize("world,kind,nx,ny,solver,entry,maxiter", [
    (2, "poisson", 96, 64, "cg", "solver", -1), (3, "random_spd", 80, 77, "cg", "module_a", -1),
    (2, "convdiff", 96, 64, "bicgstab", "solver", -1), (3, "convdiff", 96, 64, "bicgstab", "module_a", 9),
    (2, "convdiff", 96, 64, "gmres", "solver", -1), (2, "random_spd", 80, 77, "gmres_incremental", "module_a", -1),
    (2, "poisson", 4, 8000, "cg", "solver", -1)])
def test_row_partitioned_solvers_through_the_call_surface_shared_gpu(world, kind, nx, ny, solver, entry, maxiter, tmp_path):
    r = _run(world, kind, nx, ny, 1e-8, maxiter, tmp_path, "hip", solver, entry)
    _check(r, entry, "gmres" if solver.startswith("gmres") else solver)
