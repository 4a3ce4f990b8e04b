#!/usr/bin/env python3
"""bench.py -- headline benchmark of the Module-A hot path on MI355X.

Metric (BASELINE.json): CG iterations/s (+ SpMV GB/s) in fp64 on the 2-D 5-point Poisson matrix.
A STEP is one complete `cg(A, b, tol=1e-6)` solve through the public API (reference call
surface, TSL:1019) on synthetic input already resident in HBM:
  * N = 1: BASELINE config 2 -- nx = ny = 2000, N = 4,000,000 rows, nnz = 19,992,000, b = ones.
  * N > 1: the row-partitioned solver (one rank per GPU, RCCL halo exchange + partial-sum
    all-gather); weak scaling: every rank owns a 2000 x 2000 slab (grid (2000 N) x 2000), so the
    per-GPU work is fixed.  `value` = (ranks x iterations) / time = 4M-row CG iterations/s.
One JSON line is printed by rank 0.  Extra objects: `kernels` (the three kernels of the CG iteration,
HIP events around their launches inside the solver loop, algorithmic bytes), `roofline` (the
longest of them), `spmv` (the SpMV GB/s of the metric; the Poisson matrix takes the coded path --
one byte per entry -- so the general CSR kernels are measured beside it on the same matrix) and
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
PMC_FILE = "r01j_pmc_kernels.json"   # per-kernel HBM traffic from the committed rocprofv3 --pmc passes


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
    # HIPK_BENCH_DIST=1 runs the row-partitioned code path at world size 1 as well (rehearsal of the N > 1 branch on a
    # one-GPU box: same classes, same RCCL calls, no peers)
    use_dist = world > 1 or os.environ.get("HIPK_BENCH_DIST") == "1"
    json_fd = None
    if use_dist:
        # RCCL prints a version banner to STDOUT when a communicator is created; the contract is ONE JSON line there.
        # Everything but the final line goes to stderr: fd 1 is pointed at fd 2 until the result is printed.
        sys.stdout.flush()
        json_fd = os.dup(1)
        os.dup2(2, 1)
    if use_dist:
        import torch.distributed as dist
        from pytorch_sparse_solver.distributed import DistPoissonProblem, dist_cg
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29581")
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
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
    if use_dist:
        import torch.distributed as dist
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = tt.item()
    x, info, st = last

    # ---- roofline leg: each kernel of the CG iteration timed with HIP events around its launches INSIDE the
    # solver loop (params.profile selects the kernel; empty-event-pair overhead subtracted), same inputs, right
    # after the timed region.  `roofline` is the dominant (longest) kernel; `kernels` lists all three.
    roof = None
    spmv_standalone = None
    kernels = None
    spmv_report = None
    if not use_dist:
        n = nx * nx
        sv = 8
        pmc = {}
        try:
            if nx == NX:
                pmc = json.load(open(os.path.join(ROOT, "profiles", PMC_FILE)))["kernels"]
        except Exception:
            pass

        def in_loop(which, handle=h):
            xx = torch.zeros_like(b)
            pst = _hipk.solve("cg", handle, b, xx, tol=args.tol, atol=0.0, maxiter=256, profile=which)
            return pst.spmv_ms_avg * 1e3, pst.spmv_profiled, pst.event_overhead_ms * 1e3

        path = h.path()
        spmv_name = {"coded": "hipk_spmv_sell_loop_kernel<double,5,true,false,true> (coded SpMV, uniform tiles from one word per tile, + fused <p,Ap> chunk partials)",
                     "tile_fast": "hipk_spmv_kernel<double,1280,true> (CSR SpMV + fused <p,Ap> tile partials)"}.get(path, path)
        legs = [("spmv", 1, spmv_name, spmv_bytes, "SURVEY 8d: nnz*12 + (n+1)*4 + 2n*8"),
                ("cg_update", 2, "hipk_cg_update_kernel<double> (r -= alpha Ap, <r,r> partials)", 3 * n * sv,
                 "read Ap, r; write r = 24 n"),
                ("cg_direction", 3, "hipk_cg_direction_kernel<double> (x += alpha p, p = r + beta p)", 5 * n * sv,
                 "read r, p, x; write p, x = 40 n")]
        kernels = []
        for key, which, name, nbytes, what in legs:
            us, cnt, over = in_loop(which)
            kernels.append({"key": key, "kernel": name, "avg_launch_us": us, "launches_timed": cnt,
                            "algorithmic_bytes_per_launch": nbytes, "bytes_are": what,
                            "achieved_GBps": nbytes / us / 1e3, "frac_of_hbm_peak": nbytes / us / 1e3 / HBM_PEAK_GBPS,
                            "event_pair_overhead_us_subtracted": over,
                            "traffic": (pmc.get(key) or {}).get("traffic_bytes_per_launch")})
        dom = max(kernels, key=lambda k: k["avg_launch_us"])
        roof = {"bound": "hbm", "kernel": dom["kernel"] + ", timed inside the CG loop", "achieved": dom["achieved_GBps"],
                "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": dom["frac_of_hbm_peak"], "traffic": dom["traffic"],
                "traffic_source": f"profiles/{PMC_FILE} (2*FETCH_SIZE + WRITE_SIZE, separate rocprofv3 --pmc passes)",
                "avg_launch_us": dom["avg_launch_us"], "launches_timed": dom["launches_timed"],
                "event_pair_overhead_us_subtracted": dom["event_pair_overhead_us_subtracted"],
                "algorithmic_bytes_per_launch": dom["algorithmic_bytes_per_launch"],
                "why_this_kernel": "longest kernel of the iteration (see `kernels` for all three)"}

        def standalone(handle):
            g = torch.Generator(device=dev).manual_seed(0)
            xr = torch.randn(n, dtype=torch.float64, device=dev, generator=g)
            yr = torch.empty_like(xr)
            for _ in range(20):
                _hipk.spmv(handle, xr, out=yr)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(200):
                _hipk.spmv(handle, xr, out=yr)
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / 200 * 1e3

        us = standalone(h)
        spmv_standalone = {"us": us, "GB/s": spmv_bytes / us / 1e3, "reps": 200, "path": path}
        # SpMV GB/s of the metric: SURVEY-formula bytes over the in-loop kernel time.  On the coded path this is an
        # EFFECTIVE figure (the kernel streams format_bytes, not the CSR arrays); the general CSR kernels on the
        # same matrix are measured next to it.
        spmv_report = {"path": path, "in_loop_us": kernels[0]["avg_launch_us"],
                       "effective_GBps_on_csr_bytes": kernels[0]["achieved_GBps"],
                       "csr_algorithmic_bytes": spmv_bytes, "format_bytes_streamed": h.format_bytes(),
                       "GBps_on_format_bytes": h.format_bytes() / kernels[0]["avg_launch_us"] / 1e3}
        if path == "coded":
            h.set_path(plain_only=True)
            try:
                pus, pcnt, _ = in_loop(1)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                _, _, pst2 = one_solve()
                torch.cuda.synchronize()
                pdt = time.perf_counter() - t1
                spmv_report["plain_csr_kernels_same_matrix"] = {
                    "path": h.path(), "in_loop_us": pus, "GBps": spmv_bytes / pus / 1e3,
                    "frac_of_hbm_peak": spmv_bytes / pus / 1e3 / HBM_PEAK_GBPS, "standalone_us": standalone(h),
                    "cg_iters_per_sec": pst2.iterations / pdt,
                    "traffic": (pmc.get("spmv_plain") or {}).get("traffic_bytes_per_launch")}
            finally:
                h.set_path(plain_only=False)

    if use_dist:
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
        L = _hipk.lib()
        lpath = {0: "tile_fast", 1: "tile", 2: "rowwave", 3: "coded", 4: "offset_coded"}.get(int(L.hipk_csr_spmv_path(prob.A["h"])), "?")
        fbytes = int(L.hipk_csr_format_bytes(prob.A["h"]))   # bytes the selected path streams (= CSR formula unless coded)
        ach = fbytes / (ms * 1e-3) / 1e9
        roof = {"bound": "hbm", "kernel": f"SpMV of rank 0's row block ({lpath} path), stand-alone; bytes = what that path streams",
                "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS, "traffic": None,
                "avg_launch_us": ms * 1e3, "launches_timed": 100, "algorithmic_bytes_per_launch": fbytes,
                "csr_formula_bytes": prob.spmv_bytes, "effective_GBps_on_csr_bytes": prob.spmv_bytes / (ms * 1e-3) / 1e9}

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
            "spmv": spmv_report,
            "kernels": kernels,
            "roofline": roof,
        }
        if not args.no_cpu_baseline and not use_dist:
            out["cpu_baseline"] = cpu_baseline(nx, args.cpu_iters)
        line = json.dumps(out)
        if json_fd is not None:
            sys.stdout.flush()
            os.write(json_fd, (line + "\n").encode())
        else:
            print(line)
    if use_dist:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
