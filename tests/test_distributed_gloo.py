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


def _run(world, kind, nx, ny, tol, maxiter, tmp_path, mode="cpu", solver="cg"):
    out = str(tmp_path / f"res_{world}_{kind}.json")
    for _attempt in range(3):   # a port found free can be taken before the store binds it (EADDRINUSE): try another one
        port = _free_port()
        procs = []
        for r in range(world):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port), OMP_NUM_THREADS="1")
            procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_dist_worker.py"), kind, str(nx), str(ny),
                                           str(tol), str(maxiter), out, mode, solver], env=env, stdout=subprocess.PIPE,
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
        if all(p.returncode == 0 for p in procs) or not any("EADDRINUSE" in lg for lg in logs):
            break
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    with open(out) as f:
        return json.load(f)


@pytest.mark.parametrize("world,kind,nx,ny", [
    (2, "poisson", 96, 64),       # 6144 rows = 3 chunks: ranks own 2 + 1
    (3, "poisson", 96, 64),       # one chunk per rank
    (4, "poisson", 96, 64),       # rank 3 owns NO rows
    (2, "random_spd", 80, 77),    # 6160 rows (ragged last chunk), ghosts from arbitrary owners
    (3, "random_spd", 80, 77),
    (2, "poisson", 4, 8000),      # BASELINE config 5's slab shape: 8000-wide grid lines, the halo is a whole grid line
    (3, "poisson", 5, 8000),      #   and the block boundaries (chunk aligned) fall INSIDE grid lines
])
def test_dist_cg_bitwise_equals_single_rank(world, kind, nx, ny, tmp_path):
    r = _run(world, kind, nx, ny, 1e-8, -1, tmp_path)
    assert r["bitwise_equal"], r
    assert set(r["info"]) == {r["ref_info"]} == {0}
    assert set(r["iterations"]) == {r["ref_iterations"]}
    assert set(r["residual_norm"]) == {r["ref_residual_norm"]}
    assert sum(r["n_local"]) == nx * ny and r["chunk"] == 2048
    if ny == 8000:
        assert r["n_ghost"] == 8000     # rank 0 needs exactly one grid line of its neighbour


def test_dist_cg_maxiter_cutoff(tmp_path):
    r = _run(2, "poisson", 96, 64, 1e-12, 9, tmp_path)
    assert r["bitwise_equal"] and set(r["iterations"]) == {9} and set(r["info"]) == {-1} and r["ref_info"] == -1


@pytest.mark.gpu
@pytest.mark.parametrize("world,kind,nx,ny", [(2, "poisson", 96, 64), (3, "random_spd", 80, 77), (4, "poisson", 96, 64),
                                              (2, "poisson", 4, 8000)])
def test_dist_cg_hip_kernels_multi_rank_on_one_gpu(world, kind, nx, ny, tmp_path):
    """The REAL step API (libhipk.so) under a multi-rank partition: ranks share cuda:0, collectives are staged
    through the host over gloo.  Must equal the single-rank oracle solve bit for bit."""
    r = _run(world, kind, nx, ny, 1e-8, -1, tmp_path, mode="hip")
    assert r["bitwise_equal"], r
    assert set(r["info"]) == {0} and set(r["iterations"]) == {r["ref_iterations"]}


@pytest.mark.gpu
@pytest.mark.parametrize("world,kind,nx,ny,mode", [(2, "poisson", 96, 64, "native"), (2, "random_spd", 80, 77, "native"),
                                                   (2, "random_spd", 80, 77, "native_ag"), (3, "poisson", 96, 64, "native"),
                                                   (3, "poisson", 96, 64, "native_ag"), (2, "poisson", 4, 8000, "native"),
                                                   (2, "random_spd", 80, 77, "native_side"), (3, "poisson", 96, 64, "native_side")])
def test_dist_cg_c_driven_loop_multi_rank_on_one_gpu(world, kind, nx, ny, mode, tmp_path):
    """hipk_dist_cg_solve (the loop of a rank in C: fixed batches, stop word read one batch late, halo by neighbour
    send/recv pairs or by all-gathered slabs; `native_side`: x += alpha p on a side stream beside the second collective,
    HIPK_DIST_OVERLAP=1) under a multi-rank partition: ranks share cuda:0, the collective entry points
    are host-staged stand-ins (tests/_dist_worker.py).  Bitwise equal to the single-rank oracle solve."""
    r = _run(world, kind, nx, ny, 1e-8, -1, tmp_path, mode=mode)
    assert r["bitwise_equal"], r
    assert set(r["info"]) == {0} and set(r["iterations"]) == {r["ref_iterations"]}
    assert set(r["residual_norm"]) == {r["ref_residual_norm"]}


@pytest.mark.gpu
@pytest.mark.parametrize("world,kind,nx,ny", [(2, "poisson", 96, 64), (3, "poisson", 96, 64), (2, "random_spd", 80, 77),
                                              (2, "poisson", 4, 8000)])
def test_dist_cg_c_driven_loop_device_mailboxes(world, kind, nx, ny, tmp_path):
    """The same loop with the experimental device-mailbox exchange (csrc/hipk_p2p.hip, HIPK_DIST_COMM=p2p): ranks share
    cuda:0 and map each other's mailboxes through HIP IPC, so NOTHING in the iteration is staged through the host --
    the exchange kernels of different processes really wait on each other's flags.  Bitwise equal to the oracle."""
    r = _run(world, kind, nx, ny, 1e-8, -1, tmp_path, mode="native_p2p")
    assert r["bitwise_equal"], r
    assert set(r["info"]) == {0} and set(r["iterations"]) == {r["ref_iterations"]}
    assert set(r["residual_norm"]) == {r["ref_residual_norm"]}


@pytest.mark.gpu
@pytest.mark.parametrize("world,kind,nx,ny,maxiter", [(2, "poisson", 96, 64, -1), (3, "poisson", 96, 64, -1), (2, "random_spd", 80, 77, -1),
                                                      (3, "random_spd", 96, 64, -1), (2, "poisson", 4, 8000, -1), (3, "poisson", 5, 8000, -1),
                                                      (2, "poisson", 96, 64, 9)])
def test_dist_cg_exchanges_fused_into_the_kernels(world, kind, nx, ny, maxiter, tmp_path):
    """no collective launch inside the CG loop -- the update / direction kernels publish this rank's partials
    (and the boundary entries of r) into the peers' IPC-mapped mailboxes and wait for the peers' (csrc/hipk_fx.h).  Ranks share
    cuda:0, so kernels of different processes really wait on each other's stores.  One chunk per rank (grid smaller than the world:
    surplus publisher workgroups), scattered ghosts from several owners, 8000-wide grid lines with block boundaries inside them, a
    maxiter cut-off.  Bitwise equal to the single-rank oracle solve."""
    r = _run(world, kind, nx, ny, 1e-8 if maxiter < 0 else 1e-12, maxiter, tmp_path, mode="native_fused")
    assert r["bitwise_equal"], r
    assert set(r["info"]) == {r["ref_info"]} and set(r["iterations"]) == {r["ref_iterations"]}
    assert set(r["residual_norm"]) == {r["ref_residual_norm"]}


@pytest.mark.gpu
def test_dist_cg_fused_exchanges_on_large_row_blocks(tmp_path):
    """Row blocks of the size a GPU really gets (1536 x 1500 grid, 1.15 M rows per rank = 563 chunks, two ranks sharing cuda:0):
    the fused update / direction kernels as full grids, 40 iterations, bitwise equal to the single-rank oracle solve."""
    r = _run(2, "poisson", 1536, 1500, 1e-12, 40, tmp_path, mode="native_fused")
    assert r["bitwise_equal"], {k: v for k, v in r.items() if k != "residual_norm"}
    assert set(r["iterations"]) == {40} == {r["ref_iterations"]}


@pytest.mark.gpu
@pytest.mark.parametrize("world,kind,nx,ny,mode,maxiter", [(2, "convdiff", 96, 64, "native", -1), (3, "convdiff", 96, 64, "native_ag", -1),
                                                           (2, "random_spd", 80, 77, "native", -1), (2, "convdiff", 4, 8000, "native", -1),
                                                           (2, "convdiff", 96, 64, "native_p2p", -1), (2, "convdiff", 96, 64, "native", 9)])
def test_dist_bicgstab_c_driven_loop_multi_rank_on_one_gpu(world, kind, nx, ny, mode, maxiter, tmp_path):
    """hipk_dist_bicgstab_solve (row-partitioned BiCGStab, the loop of a rank in C: five collective launches per iteration) under a
    multi-rank partition sharing cuda:0 -- host-staged stand-ins for the collectives (both halo forms) or the device mailboxes.
    Bitwise equal to the single-rank oracle solve: x, iteration count, info, true residual."""
    r = _run(world, kind, nx, ny, 1e-8, maxiter, tmp_path, mode=mode, solver="bicgstab")
    assert r["bitwise_equal"], r
    assert set(r["info"]) == {r["ref_info"]} and set(r["iterations"]) == {r["ref_iterations"]}
    assert set(r["residual_norm"]) == {r["ref_residual_norm"]}
    if maxiter > 0:
        assert set(r["iterations"]) == {maxiter}


@pytest.mark.gpu
@pytest.mark.parametrize("world,kind,nx,ny,mode,solver,maxiter", [
    (2, "convdiff", 96, 64, "native", "gmres", -1), (3, "convdiff", 96, 64, "native_ag", "gmres", -1),
    (2, "random_spd", 80, 77, "native", "gmres_incremental", -1), (2, "convdiff", 4, 8000, "native", "gmres", 3),
    (2, "convdiff", 96, 64, "native_p2p", "gmres_incremental", -1)])
def test_dist_gmres_c_driven_loop_multi_rank_on_one_gpu(world, kind, nx, ny, mode, solver, maxiter, tmp_path):
    """hipk_dist_gmres_solve (row-partitioned GMRES(12): the large-system kernels with in-place all-gathers of their chunk partials
    and the halo of v_k before each SpMV) under a multi-rank partition sharing cuda:0.  Bitwise equal to the single-rank oracle
    solve: x, restart-cycle count, info, true residual."""
    r = _run(world, kind, nx, ny, 1e-8, maxiter, tmp_path, mode=mode, solver=solver)
    assert r["bitwise_equal"], r
    assert set(r["info"]) == {r["ref_info"]} and set(r["iterations"]) == {r["ref_iterations"]}
    assert set(r["residual_norm"]) == {r["ref_residual_norm"]}


@pytest.mark.gpu
def test_dist_cg_c_driven_loop_maxiter_cutoff(tmp_path):
    r = _run(2, "poisson", 96, 64, 1e-12, 9, tmp_path, mode="native")
    assert r["bitwise_equal"] and set(r["iterations"]) == {9} and set(r["info"]) == {-1} and r["ref_info"] == -1


@pytest.mark.gpu
def test_dist_cg_nccl_world1_equals_single_gpu(tmp_path):
    """RCCL path smoke test at world_size 1 (the only size the 1-GPU box allows): DistPoissonProblem + dist_cg
    through torch.distributed 'nccl' must equal the single-device cg() bit for bit."""
    code = r'''
import os, sys, json, torch, torch.distributed as dist
sys.path[:0] = [%r, %r]
from pytorch_sparse_solver.distributed import DistPoissonProblem, dist_bicgstab, dist_cg, dist_gmres
from pytorch_sparse_solver.module_a import bicgstab, cg, gmres, get_last_stats
from pytorch_sparse_solver.utils.matrix_utils import create_poisson_2d_csr
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
prob = DistPoissonProblem(nx_per_rank=96, ny=64, rank=0, world=1, device=torch.device("cuda", 0))
assert prob.comm is not None, "direct RCCL communicator expected on the nccl backend"
from pytorch_sparse_solver.distributed import native_loop_ok
assert native_loop_ok(prob), "the C-driven loop (hipk_dist_cg_solve) is the default with direct RCCL"
# exercise the grouped send/recv wrapper (a rank never sends to itself in the solver): self exchange of 3 + 2 doubles
a = torch.arange(5, dtype=torch.float64, device="cuda:0"); r = torch.zeros(5, dtype=torch.float64, device="cuda:0")
prob.comm.all_to_all(r, a, [5], [5]); torch.cuda.synchronize(); assert torch.equal(r, a)
x, info, st = dist_cg(prob, tol=1e-8)
A = create_poisson_2d_csr(96, 64, device="cuda:0")
xr, info_r = cg(A, torch.ones(96 * 64, dtype=torch.float64, device="cuda:0"), tol=1e-8)
s = get_last_stats()
xb, info_b, stb = dist_bicgstab(prob, tol=1e-8)          # the row-partitioned BiCGStab through real RCCL calls
xbr, info_br = bicgstab(A, torch.ones(96 * 64, dtype=torch.float64, device="cuda:0"), tol=1e-8)
sb = get_last_stats()
xg, info_g, stg = dist_gmres(prob, tol=1e-8, restart=15)   # and the row-partitioned GMRES
xgr, info_gr = gmres(A, torch.ones(96 * 64, dtype=torch.float64, device="cuda:0"), tol=1e-8, restart=15)
sg = get_last_stats()
print(json.dumps({"gm_equal": bool(torch.equal(xg, xgr)), "gm_info": [info_g, info_gr], "gm_it": [stg.iterations, sg.iterations],
                  "equal": bool(torch.equal(x, xr)), "info": info, "info_r": info_r, "it": st.iterations, "it_r": s.iterations,
                  "res": st.residual_norm, "res_r": s.residual_norm,
                  "bi_equal": bool(torch.equal(xb, xbr)), "bi_info": [info_b, info_br], "bi_it": [stb.iterations, sb.iterations]}))
dist.destroy_process_group()
''' % (os.path.dirname(HERE), os.path.join(os.path.dirname(HERE), "pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd"))
    for _ in range(3):     # the port found free can be taken by the time the store binds it (EADDRINUSE): take another one
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
        p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        if p.returncode == 0 or "EADDRINUSE" not in p.stderr:
            break
    assert p.returncode == 0, p.stdout + p.stderr
    r = json.loads(p.stdout.strip().splitlines()[-1])
    assert r["equal"] and r["info"] == r["info_r"] == 0 and r["it"] == r["it_r"] and r["res"] == r["res_r"], r
    assert r["bi_equal"] and r["bi_info"] == [0, 0] and r["bi_it"][0] == r["bi_it"][1], r
    assert r["gm_equal"] and r["gm_info"][0] == r["gm_info"][1] and r["gm_it"][0] == r["gm_it"][1], r


@pytest.mark.gpu
@pytest.mark.parametrize("strided", ["0", "1"])
@pytest.mark.parametrize("kind,solver,maxiter", [("poisson", "cg", 20), ("convdiff", "bicgstab", 10), ("convdiff", "gmres", 1)])
def test_dist_large_row_blocks_take_the_two_rows_per_lane_kernel(kind, solver, maxiter, strided, tmp_path, monkeypatch):
    """Row blocks of the size a GPU really gets (1536 x 1500 grid, 1.15 M rows per rank, two ranks sharing cuda:0, host-staged
    collectives): the local matrices -- square interior plus halo columns -- take the coded SpMV's two-rows-per-lane kernel, with
    the chunk walk (HIPK_SPMV_SELL_STRIDED=0) and with the grouped walk that a rank of a 4- / 8-rank run takes (=1; tile sums
    through the combine kernel into the rank's slice of the global partials).  A few iterations of each row-partitioned solver,
    bitwise equal to the single-rank oracle solve."""
    monkeypatch.setenv("HIPK_SPMV_SELL_STRIDED", strided)
    r = _run(2, kind, 1536, 1500, 1e-12, maxiter, tmp_path, mode="native", solver=solver)
    assert r["bitwise_equal"], {k: v for k, v in r.items() if k != "residual_norm"}
    assert set(r["info"]) == {r["ref_info"]} and set(r["iterations"]) == {r["ref_iterations"]}
    assert set(r["residual_norm"]) == {r["ref_residual_norm"]}
    for k in r["spmv_kernel"]:
        assert k.startswith("hipk_spmv_sell_wide_kernel") and k.endswith("," + strided + ">"), r["spmv_kernel"]
