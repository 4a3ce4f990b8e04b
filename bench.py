#!/usr/bin/env python3
"""bench.py -- headline benchmark of the Module-A hot path on MI355X.

Metric (BASELINE.json): CG iterations/s (+ SpMV GB/s) in fp64 on the 2-D 5-point Poisson matrix.
A STEP is one complete `cg(A, b, tol=1e-6)` solve through the public API (reference call
surface, TSL:1019) on synthetic input already resident in HBM:
  * N = 1: BASELINE config 2 -- nx = ny = 2000, N = 4,000,000 rows, nnz = 19,992,000, b = ones.
  * N > 1: the row-partitioned solver (one rank per GPU, RCCL halo exchange + partial-sum
    all-gather); weak scaling: every rank owns a 2000 x 2000 slab (grid (2000 N) x 2000), so the
    per-GPU work is fixed.  `value` = (ranks x iterations) / time = 4M-row CG iterations/s.
One JSON line is printed by rank 0.  Extra objects: `roofline` (fused SpMV+dot kernel, HIP
events around its launches inside the solver loop, algorithmic bytes of SURVEY 8d) and
`cpu_baseline` (the oracle's C restatement on the host cores, bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
NX = 2000


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--nx", type=int, default=NX, help="grid lines per GPU (default 2000 = BASELINE config 2)")
    ap.add_argument("--tol", type=float, default=1e-6)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=6000)
    return ap.parse_args()


def host_cores():
    """Threads the CPU baseline may use: the cgroup CPU quota if there is one, else the affinity mask,
    capped at 64 (the oracle's loops stop scaling long before)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 64))


def cpu_baseline(nx, iters):
    """Oracle CG (oracle/krylov_oracle.c, OpenMP over rows/chunks) on the host cores: bounded sample."""
    import numpy as np
    from oracle import oracle as O
    from pytorch_sparse_solver.utils.matrix_utils import stencil5_csr_components
    O.build()
    cores = host_cores()
    O.set_threads(cores)
    crow, col, val = stencil5_csr_components(nx, nx, 4.0, -1.0, -1.0, -1.0, -1.0, index_dtype=torch.int32)
    crow, col, val = crow.numpy(), col.numpy(), val.numpy()
    b = np.ones(nx * nx)
    O.cg(crow, col, val, b, tol=0.0, maxiter=5)  # warm up threads / page in
    t0 = time.perf_counter()
    r = O.cg(crow, col, val, b, tol=0.0, maxiter=iters)
    dt = time.perf_counter() - t0
    x = np.random.default_rng(0).standard_normal(nx * nx)
    t1 = time.perf_counter()
    for _ in range(20):
        O.spmv(crow, col, val, x)
    spmv_s = (time.perf_counter() - t1) / 20
    nnz = int(crow[-1])
    bytes_spmv = nnz * 12 + (nx * nx + 1) * 4 + 2 * nx * nx * 8
    O.set_threads(1)
    return {"value": r.iterations / dt, "unit": "it/s", "cores": cores, "kind": "port",
            "sample": f"oracle CG, same N={nx * nx} Poisson matrix and b=ones, {r.iterations} iterations "
                      f"({dt:.1f} s incl. 2 residual SpMVs); oracle SpMV {spmv_s * 1e3:.2f} ms = "
                      f"{bytes_spmv / spmv_s / 1e9:.1f} GB/s"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with torch.distributed.run (one rank per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from pytorch_sparse_solver import _hipk
    from pytorch_sparse_solver.module_a import cg, get_last_stats
    from pytorch_sparse_solver.utils.matrix_utils import create_poisson_2d_csr

    nx = args.nx
    if world > 1:
        import torch.distributed as dist
        from pytorch_sparse_solver.distributed import DistPoissonProblem, dist_cg
        dist.init_process_group("nccl", device_id=dev)
        prob = DistPoissonProblem(nx_per_rank=nx, ny=nx, rank=rank, world=world, device=dev)

        def barrier():
            dist.barrier()

        def one_solve():
            return dist_cg(prob, tol=args.tol)
        n_rows_rank, nnz_rank, spmv_bytes = prob.n_local, prob.nnz_local, prob.spmv_bytes
        workload = f"poisson5pt_{nx * world}x{nx}_rowpart_{world}ranks_cg_tol{args.tol:g}_b=ones"
    else:
        A = create_poisson_2d_csr(nx, nx, device=dev)
        b = torch.ones(nx * nx, dtype=torch.float64, device=dev)
        h = _hipk.handle_for(A)

        def barrier():
            pass

        def one_solve():
            x, info = cg(A, b, tol=args.tol)
            st = get_last_stats()
            return x, info, st
        n_rows_rank, nnz_rank, spmv_bytes = nx * nx, h.nnz, h.spmv_bytes()
        workload = f"poisson5pt_{nx}x{nx}_cg_tol{args.tol:g}_b=ones"

    for _ in range(args.warmup):
        one_solve()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    iters_total, last = 0, None
    for _ in range(args.steps):
        last = one_solve()
        iters_total += last[2].iterations
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = tt.item()
    x, info, st = last

    # ---- roofline leg: the fused SpMV+dot kernel timed with HIP events around its launches
    # inside the solver loop (params.profile), same inputs, right after the timed region.
    roof = None
    spmv_standalone = None
    if world == 1:
        xx = torch.zeros_like(b)
        pst = _hipk.solve("cg", h, b, xx, tol=args.tol, atol=0.0, maxiter=256, profile=True)
        ach = spmv_bytes / (pst.spmv_ms_avg * 1e-3) / 1e9
        traffic = None   # HBM bytes per launch from the committed PMC passes (collected separately with rocprofv3 --pmc)
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01g_pmc_spmv.json")))
            if nx == NX:
                traffic = pmc["traffic_bytes_per_launch"]
        except Exception:
            pass
        roof = {"bound": "hbm", "kernel": "hipk_spmv_kernel<double,1280,true> (CSR SpMV + fused <p,Ap> tile partials), "
                                          "timed inside the CG loop",
                "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS,
                "traffic": traffic, "traffic_source": "profiles/r01g_pmc_spmv.json (2*FETCH_SIZE + WRITE_SIZE)",
                "avg_launch_us": pst.spmv_ms_avg * 1e3, "launches_timed": pst.spmv_profiled,
                "event_pair_overhead_us_subtracted": pst.event_overhead_ms * 1e3,
                "algorithmic_bytes_per_launch": spmv_bytes}
        g = torch.Generator(device=dev).manual_seed(0)
        xr = torch.randn(nx * nx, dtype=torch.float64, device=dev, generator=g)
        yr = torch.empty_like(xr)
        for _ in range(20):
            _hipk.spmv(h, xr, out=yr)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            _hipk.spmv(h, xr, out=yr)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 200
        spmv_standalone = {"us": ms * 1e3, "GB/s": spmv_bytes / (ms * 1e-3) / 1e9, "reps": 200}

    if world > 1:
        # roofline leg at N > 1: this rank's local SpMV (no communication), HIP events on the launch stream
        xe = prob.ops.zeros(max(prob.n_ext, 1))
        xe.normal_(generator=torch.Generator(device=dev).manual_seed(rank))
        ye = prob.ops.zeros(prob.n_local)
        for _ in range(10):
            prob.ops.spmv(prob.A, xe, ye)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            prob.ops.spmv(prob.A, xe, ye)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 100
        ach = prob.spmv_bytes / (ms * 1e-3) / 1e9
        roof = {"bound": "hbm", "kernel": "hipk_spmv_kernel<double,1280,*> on rank 0's row block, stand-alone",
                "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS, "traffic": None,
                "avg_launch_us": ms * 1e3, "launches_timed": 100, "algorithmic_bytes_per_launch": prob.spmv_bytes}

    if rank == 0:
        out = {
            "metric": "cg_iters_per_sec",
            "value": world * iters_total / dt,
            "unit": "it/s (4M-row 5-pt Poisson CG iterations, summed over ranks)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload, "rows_per_gpu": n_rows_rank, "nnz_per_gpu": nnz_rank,
                       "iterations_per_solve": st.iterations, "info": info,
                       "relres": st.residual_norm / st.b_norm, "step": "one full cg() solve via the public API"},
            "spmv_standalone": spmv_standalone,
            "roofline": roof,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(nx, args.cpu_iters)
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
