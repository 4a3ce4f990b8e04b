"""Row-partitioned CG (pytorch_sparse_solver.distributed) with world_size > 1 on CPU: gloo backend,
the ops test double of tests/dist_cpu_ops.py (same arithmetic spec as the HIP kernels).  Checks the
partition / halo plan / collective order, and that the solve is BITWISE identical to the single-rank
oracle solve for every rank count (including a rank that owns no rows)."""
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


def _run(world, kind, nx, ny, tol, maxiter, tmp_path):
    out = str(tmp_path / f"res_{world}_{kind}.json")
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_dist_worker.py"), kind, str(nx), str(ny),
                                       str(tol), str(maxiter), out], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o.decode(errors="replace"))
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    with open(out) as f:
        return json.load(f)


@pytest.mark.parametrize("world,kind,nx,ny", [
    (2, "poisson", 96, 64),       # 6144 rows = 3 chunks: ranks own 2 + 1
    (3, "poisson", 96, 64),       # one chunk per rank
    (4, "poisson", 96, 64),       # rank 3 owns NO rows
    (2, "random_spd", 80, 77),    # 6160 rows (ragged last chunk), ghosts from arbitrary owners
    (3, "random_spd", 80, 77),
])
def test_dist_cg_bitwise_equals_single_rank(world, kind, nx, ny, tmp_path):
    r = _run(world, kind, nx, ny, 1e-8, -1, tmp_path)
    assert r["bitwise_equal"], r
    assert set(r["info"]) == {r["ref_info"]} == {0}
    assert set(r["iterations"]) == {r["ref_iterations"]}
    assert set(r["residual_norm"]) == {r["ref_residual_norm"]}
    assert sum(r["n_local"]) == nx * ny and r["chunk"] == 2048


def test_dist_cg_maxiter_cutoff(tmp_path):
    r = _run(2, "poisson", 96, 64, 1e-12, 9, tmp_path)
    assert r["bitwise_equal"] and set(r["iterations"]) == {9} and set(r["info"]) == {-1} and r["ref_info"] == -1
