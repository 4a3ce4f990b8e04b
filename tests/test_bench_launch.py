"""bench.py launch contract: `python bench.py --gpus N` as a plain process starts its N ranks itself as
CHILD processes and relays rank 0's JSON line; a failing child fails the parent."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None, timeout=600):
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None), e.pop("RANK", None), e.pop("LOCAL_RANK", None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True,
                          timeout=timeout, env=e)


def test_self_launch_spawns_ranks_and_propagates_failure():
    """No GPU here: both child ranks must come up under torch.distributed.run, refuse loudly ("needs a GPU"), and the
    parent must exit non-zero without printing a JSON line -- i.e. the self-launch path works and cannot report success
    for a failed run."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("CPU-only check")
    p = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--nx", "64"], timeout=300)
    assert p.returncode != 0
    assert "needs a GPU" in p.stderr and "2-rank child run failed" in p.stderr
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]


@pytest.mark.gpu
def test_bench_line_default_and_dist_rehearsal():
    """Small grids: the N = 1 line (roofline + kernels; cpu_baseline skipped off the headline size) and the row-partitioned
    branch at world 1 through RCCL in both scaling modes."""
    p = _run(["--steps", "1", "--warmup", "1", "--nx", "256", "--no-cpu-baseline"])
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads(p.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 1 and d["scaling"] == "weak" and d["config"]["info"] == 0 and d["value"] > 0
    roof = d["roofline"]
    assert roof["bound"] == "infinity-cache+hbm" and 0 < roof["frac"] < 1.2
    assert "nothing subtracted" in roof["timing"] and len(roof["library_build_id"]) == 16
    # legs: the dominant CG kernel and the general CSR SpMV (the N = 64 M leg only runs at the headline size)
    assert roof["legs"]["cg_dominant_n4m"]["frac"] == roof["frac"]
    assert roof["legs"]["csr_spmv_n4m"]["bound"] == "hbm" and roof["legs"]["csr_spmv_n4m"]["avg_launch_us"] > 0
    assert "hipk_spmv_kernel" in roof["legs"]["csr_spmv_n4m"]["kernel"]
    # a traffic figure is only quoted from a counter profile of the library build that is running
    assert roof["traffic"] is None or roof["traffic_from_build"] == roof["library_build_id"]
    assert d["config"]["cold_first_solve_ms"] > d["config"]["handle_creation_ms_outside_timed_region"] > 0
    for k in d["kernels"]:
        assert 0 < k["frac_of_hbm_peak"] < 1.2, k       # no fraction-of-peak above what a cache-resident toy grid can show
    assert d["spmv"]["GBps_on_format_bytes"] > 0 and "effective_GBps_on_csr_bytes" in d["spmv"]
    for extra in ([], ["--scaling", "strong", "--global-nx", "256"]):
        p = _run(["--steps", "1", "--warmup", "0", "--nx", "256", "--no-cpu-baseline"] + extra, env={"HIPK_BENCH_DIST": "1"})
        assert p.returncode == 0, p.stderr[-2000:]
        lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1                           # RCCL's banner must not reach stdout
        r = json.loads(lines[0])
        assert r["config"]["info"] == 0 and r["config"]["iterations_per_solve"] == d["config"]["iterations_per_solve"]
        assert r["scaling"] == ("strong" if extra else "weak") and r["roofline"]["frac"] > 0
