#!/usr/bin/env python3
"""bench.py -- headline benchmark of the Module-A hot path on MI355X.

Metric (BASELINE.json): CG iterations/s (+ SpMV GB/s) in fp64 on the 2-D 5-point Poisson matrix.
A STEP is one complete `cg(A, b, tol=1e-6)` solve (reference call surface, TSL:1019) on synthetic input already
resident in HBM.  Workloads:
  * N = 1 (default): BASELINE config 2 -- nx = ny = 2000, N = 4,000,000 rows, nnz = 19,992,000, b = ones.
  * N > 1, `--scaling weak` (default): the row-partitioned solver (one rank per GPU, RCCL: all-gather of the dot
    partials + the x-vector halo); every rank owns a 2000 x 2000 slab of a (2000 N) x 2000 grid, so the per-GPU work is
    config 2's.  `value` = (ranks x iterations) / time = 4M-row CG iterations/s summed over the ranks.
  * `--scaling strong [--global-nx 8000]`: BASELINE config 5 -- ONE 8000 x 8000 Poisson system (N = 64 M rows) split
    over the N ranks (N = 1: the single-device `cg`; HIPK_BENCH_DIST=1 runs the row-partitioned code at world 1).
    `value` = iterations / time of that fixed problem.
Launch: `python bench.py --gpus N ...` starts its N ranks itself (children of `python -m torch.distributed.run`,
spawned BEFORE this process touches the GPU; the JSON line is relayed, a failing child fails the parent); under
`torch.distributed.run` (WORLD_SIZE set) it is a rank.  Rank 0 prints ONE JSON line.
Extra objects: `kernels` (the three kernels of the CG iteration, each timed INSIDE the solver loop by start/stop events bound to
its dispatch -- hipExtLaunchKernel -- i.e. the dispatch's own begin/end timestamps, the durations `rocprofv3 --kernel-trace`
prints; nothing is calibrated or subtracted), `roofline` (the longest of them, plus `legs`: that kernel, the GENERAL CSR SpMV on
SURVEY 8d's bytes -- the north star's kernel -- and the same CG kernels on the N = 64 M system, whose vectors cannot sit in the
256 MiB Infinity Cache: the HBM-resident figure), `spmv` (the metric's SpMV GB/s; the Poisson matrix takes the coded
path -- one byte per entry -- so the general CSR kernels are measured beside it on the same matrix), `cpu_baseline` (the
oracle's C restatement on the host cores + the torch-CPU generic loop, bounded samples).  `traffic` comes from committed
rocprofv3 --pmc passes and is quoted only when they were taken on the library build that is running (hipk_build_id).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
NX = 2000
# per-kernel HBM traffic from committed rocprofv3 --pmc passes (tools/prof_bench.sh + tools/pmc_to_json.py): NOT measured
# in this run (PMC needs rocprofv3 attached); stamped with the commit it was taken at and dropped when the kernel differs
PMC_FILES = ("r03_pmc_kernels.json",)
STATS_FILE = "r03_bench_kernel_stats.json"   # tools/kernel_stats_to_json.py: rocprofv3 --kernel-trace --stats of this command


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--nx", type=int, default=NX, help="weak scaling: grid lines per GPU (default 2000 = BASELINE config 2)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="strong: ONE global-nx x global-nx system split over the ranks (BASELINE config 5)")
    ap.add_argument("--global-nx", type=int, default=8000, help="strong scaling: global grid is global-nx x global-nx")
    ap.add_argument("--tol", type=float, default=1e-6)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=6000)
    ap.add_argument("--no-n64m", action="store_true", help="skip the N = 64 M (HBM-resident) roofline leg")
    return ap.parse_args()


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(args):
    """`python bench.py --gpus N` as a plain process: run the N ranks as CHILD processes (torch.distributed.run) and relay
    rank 0's JSON line.  Nothing in this parent has touched the GPU (no torch.cuda call, torch not even imported)."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "4")
    for attempt in range(3):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        sys.stderr.write(proc.stderr.decode(errors="replace"))
        # the port found free can be taken by the time the rendezvous store binds it: that one failure is retried
        if proc.returncode == 0 or b"EADDRINUSE" not in proc.stderr:
            break
    line = None
    for ln in proc.stdout.decode(errors="replace").splitlines():
        ln = ln.strip()
        if ln.startswith("{") and ln.endswith("}"):
            try:
                json.loads(ln)
                line = ln
            except ValueError:
                pass
    if proc.returncode != 0 or line is None:
        sys.stderr.write(f"bench.py: the {args.gpus}-rank child run failed (exit code {proc.returncode}, "
                         f"{'no ' if line is None else ''}JSON line)\n")
        sys.exit(proc.returncode if proc.returncode != 0 else 1)
    print(line)
    sys.exit(0)


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
    """Two CPU legs on the box's host cores, bounded samples of the SAME workload:
    (1) `port`: the oracle's C restatement (oracle/krylov_oracle.c, OpenMP over rows/chunks) -- `value`;
    (2) the torch-CPU generic loop (this package's generic path = the reference's algorithm in ATen ops with
        `torch.matmul(A_csr, v)`, TSL:191 -- the analogue of the reference's own CPU execution) -- `torch_generic_it_s`."""
    import numpy as np
    import torch
    from oracle import oracle as O
    from pytorch_sparse_solver.module_a import cg, get_last_stats
    from pytorch_sparse_solver.utils.matrix_utils import create_poisson_2d_csr, stencil5_csr_components
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
    # torch generic loop: int64 CSR as torch stores it, torch's own intra-op threads
    old_threads = torch.get_num_threads()
    torch.set_num_threads(cores)
    A_cpu = create_poisson_2d_csr(nx, nx)
    b_cpu = torch.ones(nx * nx, dtype=torch.float64)
    cg(A_cpu, b_cpu, tol=0.0, maxiter=3)
    n_gen = 60
    t2 = time.perf_counter()
    cg(A_cpu, b_cpu, tol=0.0, maxiter=n_gen)
    dt_gen = time.perf_counter() - t2
    it_gen = get_last_stats().iterations
    torch.set_num_threads(old_threads)
    return {"value": r.iterations / dt, "unit": "it/s", "cores": cores, "kind": "port",
            "sample": f"oracle CG (C/OpenMP restatement), same N={nx * nx} Poisson matrix and b=ones, {r.iterations} iterations "
                      f"({dt:.1f} s incl. 2 residual SpMVs); oracle SpMV {spmv_s * 1e3:.2f} ms = "
                      f"{bytes_spmv / spmv_s / 1e9:.1f} GB/s",
            "torch_generic_it_s": it_gen / dt_gen,
            "torch_generic_sample": f"generic torch-op CG loop (torch.matmul(A_csr, v), int64 CSR, {cores} intra-op threads), "
                                    f"{it_gen} iterations in {dt_gen:.2f} s"}


def _loop_spmv_kernel(h, v):
    """Name of the kernel instantiation the CG loop's SpMV (y = A p with the fused <p, y> partials) selects for this handle:
    one such launch through the C ABI, then the library's own record of what it dispatched (hipk_last_spmv_kernel)."""
    import torch
    from pytorch_sparse_solver import _hipk
    L = _hipk.lib()
    y = torch.empty_like(v)
    part = torch.zeros(int(L.hipk_chunk_count(v.numel())), dtype=torch.float64, device=v.device)
    _hipk._check(L.hipk_spmv_ex(h._h, v.data_ptr(), y.data_ptr(), 1, v.data_ptr(), None, part.data_ptr(), None, None, 0,
                                torch.cuda.current_stream().cuda_stream), "hipk_spmv_ex")
    torch.cuda.synchronize()
    return h.last_spmv_kernel()


def load_pmc(build_id):
    """Committed per-kernel counter traffic, only if it was taken on the library build that is running."""
    for name in PMC_FILES:
        try:
            d = json.load(open(os.path.join(ROOT, "profiles", name)))
        except Exception:
            continue
        if d.get("build_id") == build_id:
            return name, d, None
        return name, {}, f"profiles/{name} was taken on build {d.get('build_id')}, this library is build {build_id}"
    return None, {}, "no counter profile committed for this round"


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args)   # never returns
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from pytorch_sparse_solver import _hipk
    from pytorch_sparse_solver.module_a import cg, get_last_stats
    from pytorch_sparse_solver.utils.matrix_utils import create_poisson_2d_csr

    strong = args.scaling == "strong"
    nx = args.global_nx if strong else args.nx
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
    handle_ms = None
    cold_first_solve_ms = None
    if use_dist:
        import torch.distributed as dist
        from pytorch_sparse_solver import RowBlockCSR
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(free_port()) if world == 1 else "29581")
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        # the row-partitioned solver is reached through the reference's call surface: cg(A, b, tol=...) (TSL:1019) with this
        # rank's ROW BLOCK of the global matrix as the operand and its slice of b
        if strong:
            gx = nx
            workload = f"poisson5pt_{nx}x{nx}_N={nx * nx}_rowpart_{world}ranks_strong_cg_tol{args.tol:g}_b=ones"
        else:
            gx = nx * world
            workload = f"poisson5pt_{nx * world}x{nx}_rowpart_{world}ranks_weak_cg_tol{args.tol:g}_b=ones"
        A_rb = RowBlockCSR.poisson5(gx, nx, device=dev)
        r0, r1 = RowBlockCSR.row_range(gx * nx)
        b_loc = torch.ones(r1 - r0, dtype=torch.float64, device=dev)
        prob = A_rb.problem(b_loc)      # halo plan, device matrix, communicator (the ranks agree on it here, before any solve)

        def barrier():
            dist.barrier()

        def one_solve():
            x, info = cg(A_rb, b_loc, tol=args.tol)
            return x, info, get_last_stats()
        n_rows_rank, nnz_rank, spmv_bytes = prob.n_local, prob.nnz_local, prob.spmv_bytes
    else:
        A = create_poisson_2d_csr(nx, nx, device=dev)
        b = torch.ones(nx * nx, dtype=torch.float64, device=dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        h = _hipk.handle_for(A)     # index narrowing + validation, dictionary, code planes, uniform-tile analysis
        torch.cuda.synchronize()
        handle_ms = (time.perf_counter() - t0) * 1e3
        # the first cg() of a process on a fresh matrix, as a user pays it: handle creation (above) + the first solve
        # (code-object load of the kernels it touches included); outside the timed region, reported in config
        t0 = time.perf_counter()
        cg(A, b, tol=args.tol)
        torch.cuda.synchronize()
        cold_first_solve_ms = handle_ms + (time.perf_counter() - t0) * 1e3

        def barrier():
            pass

        def one_solve():
            x, info = cg(A, b, tol=args.tol)
            st = get_last_stats()
            return x, info, st
        n_rows_rank, nnz_rank, spmv_bytes = nx * nx, h.nnz, h.spmv_bytes()
        workload = f"poisson5pt_{nx}x{nx}_N={nx * nx}_cg_tol{args.tol:g}_b=ones" + ("_strong_1rank" if strong else "")

    for w in range(args.warmup):
        one_solve()     # a rank that fails raises: its process exits non-zero and the launcher tears the job down
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

    # ---- roofline leg: each kernel of the CG iteration timed INSIDE the solver loop by start/stop events bound to its
    # dispatch (params.profile selects the kernel; hipk_solve.h: hipExtLaunchKernel): the dispatch's own begin/end
    # timestamps, i.e. the durations rocprofv3 --kernel-trace prints for the same launches.  Nothing is subtracted.
    # `roofline` is the dominant (longest) kernel of the headline iteration; `roofline.legs` adds the general CSR SpMV and
    # the N = 64 M (HBM-resident) run of the same CG kernels; `kernels` lists all three kernels of the iteration.
    roof = None
    spmv_standalone = None
    kernels = None
    spmv_report = None
    TIMING = ("avg_launch_us = stop - start of start/stop events bound to the kernel's dispatch (hipExtLaunchKernel), averaged over "
              "the launches of 256 CG iterations inside the solver loop; RAW, nothing subtracted: the start stamp is taken when the "
              "dispatch is picked up, 0.6-1.5 us before the first wave while the previous kernel drains, so these figures are "
              "conservative by 4-9 % on the 16-25 us kernels and < 1 % on the N = 64 M ones (csrc/hipk_solve.h); rocprofv3_avg_us / "
              "frac_rocprofv3 = the average of `rocprofv3 --kernel-trace --stats` on the same command, from the committed profile of "
              "the SAME library build (profiles/" + (STATS_FILE) + "), null when the build differs")
    if not use_dist:
        n = nx * nx
        sv = 8
        build_id = _hipk.lib().hipk_build_id().decode()
        pmc_name, pmc_doc, pmc_dropped = load_pmc(build_id)
        try:
            prof_doc = json.load(open(os.path.join(ROOT, "profiles", STATS_FILE)))
        except Exception:
            prof_doc = {}
        prof_kernels = prof_doc.get("kernels", {}) if prof_doc.get("build_id") == build_id else {}

        def rocprof_us(kernel_ran):
            """rocprofv3's average duration of this kernel instantiation in the committed profile of THIS build, or None."""
            e = prof_kernels.get(kernel_ran.split(" (")[0].replace(" ", ""))
            return None if e is None else e["avg_us"]

        def traffic_of(section, key, kernel_ran):
            """PMC traffic of the committed profile of THIS build, only when it was taken on the same kernel instantiation."""
            e = pmc_doc.get(section, {}).get(key)
            if not e:
                return None
            prof_kernel = e.get("kernel", "").replace(" ", "")
            if not kernel_ran.replace(" ", "").startswith(prof_kernel.split("(")[0]):
                return None
            return e.get("traffic_bytes_per_launch")

        def in_loop(which, handle, rhs):
            xx = torch.zeros_like(rhs)
            pst = _hipk.solve("cg", handle, rhs, xx, tol=args.tol, atol=0.0, maxiter=256, profile=which)
            return pst.spmv_ms_avg * 1e3, pst.spmv_profiled

        def cg_legs(handle, rhs, nn, section, bound):
            """The three kernels of the CG iteration on `handle`: name, algorithmic bytes, in-loop duration."""
            path = handle.path()
            fb = handle.format_bytes()
            sb = handle.spmv_bytes()
            spmv_kernel = _loop_spmv_kernel(handle, rhs)
            coded_ = path in ("coded", "offset_coded")
            spmv_name = (f"{spmv_kernel} (coded SpMV + fused <p,Ap> partials)" if coded_
                         else f"{spmv_kernel} (CSR SpMV + fused <p,Ap> tile partials; {path})")
            # x, r, p, Ap beyond 384 MiB: the vector kernels' non-temporal instantiation (csrc/hipk_cg.hip)
            nt = 4 * nn * sv > 384 * 1024 * 1024
            nts = "true" if nt else "false"
            flat = nt and os.environ.get("HIPK_CG_FLAT_DIRECTION", "1" if nn * sv > 256 * 1024 * 1024 else "0")[0] != "0"
            legs = [("spmv", 1, spmv_name, fb if coded_ else sb,
                     "bytes this format streams: code planes + x + y" if coded_ else "SURVEY 8d: nnz*12 + (n+1)*4 + 2n*8"),
                    ("cg_update", 2, f"hipk_cg_update_kernel<double,false,{nts}> (r -= alpha Ap, <r,r> partials)", 3 * nn * sv,
                     "read Ap, r; write r = 24 n"),
                    ("cg_direction", 3,
                     ("hipk_cg_direction_flat_kernel<double> (x += alpha p, p = r + beta p; the one-workgroup hipk_cg_scalars_kernel "
                      "before it is timed separately: `scalars_launch_us`)" if flat else
                      f"hipk_cg_direction_kernel<double,false,{nts},false> (x += alpha p, p = r + beta p)"), 5 * nn * sv,
                     "read r, p, x; write p, x = 40 n")]
            out = []
            for key, which, name, nbytes, what in legs:
                us, cnt = in_loop(which, handle, rhs)
                pus_ = rocprof_us(name)
                k = {"key": key, "kernel": name, "bound": bound, "avg_launch_us": us, "launches_timed": cnt,
                     "rocprofv3_avg_us": pus_, "frac_rocprofv3": None if pus_ is None else nbytes / pus_ / 1e3 / HBM_PEAK_GBPS,
                     "algorithmic_bytes_per_launch": nbytes, "bytes_are": what,
                     "achieved_GBps": nbytes / us / 1e3, "frac_of_hbm_peak": nbytes / us / 1e3 / HBM_PEAK_GBPS,
                     "traffic": traffic_of(section, key, name)}
                if key == "spmv" and coded_:   # CSR-formula bytes over the coded kernel's time: an EFFECTIVE figure, no fraction of peak
                    k["effective_GBps_on_csr_bytes"] = sb / us / 1e3
                    k["csr_formula_bytes"] = sb
                if key == "cg_direction" and flat:
                    k["scalars_launch_us"] = in_loop(4, handle, rhs)[0]
                    k["avg_launch_us_is"] = "the flat-grid launch alone; the step's two launches: avg_launch_us + scalars_launch_us"
                out.append(k)
            return out

        def roof_of(k, why):
            return {"bound": k["bound"], "kernel": k["kernel"] + ", timed inside the CG loop", "achieved": k["achieved_GBps"],
                    "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": k["frac_of_hbm_peak"], "traffic": k["traffic"],
                    "avg_launch_us": k["avg_launch_us"], "launches_timed": k["launches_timed"],
                    "rocprofv3_avg_us": k["rocprofv3_avg_us"], "frac_rocprofv3": k["frac_rocprofv3"], "algorithmic_bytes_per_launch": k["algorithmic_bytes_per_launch"], "bytes_are": k["bytes_are"], "why": why}

        path = h.path()
        fbytes = h.format_bytes()
        coded = path in ("coded", "offset_coded")
        # N = 4 M: the iteration's working set on the coded path (codes 20 MB + x, r, p, Ap 128 MB) sits in the 256 MiB
        # Infinity Cache, so these kernels are fed by that cache AND HBM: their rates are not fractions of an HBM roofline
        kernels = cg_legs(h, b, n, "kernels", "infinity-cache+hbm")
        dom = max(kernels, key=lambda k: k["avg_launch_us"])
        roof = roof_of(dom, "longest kernel of the headline (N = 4 M) CG iteration")
        roof["timing"] = TIMING
        # cross-check: the iteration time of the timed region (wall clock / iterations) against the sums of the three kernels
        roof["iteration_us_from_timed_region"] = dt / max(iters_total, 1) * 1e6
        roof["iteration_us_sum_of_event_figures"] = sum(k["avg_launch_us"] for k in kernels)
        roof["iteration_us_sum_of_rocprofv3_averages"] = (sum(k["rocprofv3_avg_us"] for k in kernels)
                                                          if all(k["rocprofv3_avg_us"] is not None for k in kernels) else None)
        roof["rocprofv3_profile"] = f"profiles/{STATS_FILE}" if prof_kernels else None
        roof["traffic_from_build"] = build_id if dom["traffic"] is not None else None
        roof["traffic_source"] = (f"profiles/{pmc_name} (2*FETCH_SIZE + WRITE_SIZE, separate rocprofv3 --pmc passes, same library build)"
                                  if dom["traffic"] is not None else None)
        roof["traffic_dropped_because"] = pmc_dropped if dom["traffic"] is None else None
        roof["library_build_id"] = build_id
        roof["legs"] = {"cg_dominant_n4m": roof_of(dom, "longest kernel of the headline CG iteration (working set in the Infinity Cache)")}

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
        spmv_standalone = {"us": us, "GBps_on_format_bytes": fbytes / us / 1e3, "format_bytes": fbytes, "reps": 200,
                           "path": path}
        # SpMV GB/s of the metric.  On the coded path the kernel streams format_bytes, not the CSR arrays: its bandwidth is
        # quoted on those; the CSR-formula figure is kept as an EFFECTIVE rate only.  The general CSR kernels (the
        # north star's ">= 70 % of the HBM roofline on CSR SpMV") are measured next to it on the same matrix.
        spmv_report = {"path": path, "in_loop_us": kernels[0]["avg_launch_us"],
                       "format_bytes_streamed": fbytes, "GBps_on_format_bytes": fbytes / kernels[0]["avg_launch_us"] / 1e3,
                       "frac_on_format_bytes": fbytes / kernels[0]["avg_launch_us"] / 1e3 / HBM_PEAK_GBPS,
                       "csr_algorithmic_bytes": spmv_bytes,
                       "effective_GBps_on_csr_bytes": spmv_bytes / kernels[0]["avg_launch_us"] / 1e3}
        if coded:
            h.set_path(plain_only=True)
            try:
                pus, pcnt = in_loop(1, h, b)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                _, _, pst2 = one_solve()
                torch.cuda.synchronize()
                pdt = time.perf_counter() - t1
                pname = _loop_spmv_kernel(h, b)
                spmv_report["plain_csr_kernels_same_matrix"] = {
                    "path": h.path(), "kernel": pname, "in_loop_us": pus, "algorithmic_bytes_per_launch": spmv_bytes,
                    "GBps": spmv_bytes / pus / 1e3,
                    "frac_of_hbm_peak": spmv_bytes / pus / 1e3 / HBM_PEAK_GBPS, "standalone_us": standalone(h),
                    "cg_iters_per_sec": pst2.iterations / pdt,
                    "traffic": traffic_of("kernels", "spmv_plain", pname)}
                # the north star's kernel: general CSR SpMV on SURVEY 8d's bytes.  320 MB per product, 160 MB of it (`val`)
                # loaded non-temporal: beyond the Infinity Cache -> an HBM figure
                roof["legs"]["csr_spmv_n4m"] = {
                    "bound": "hbm", "kernel": f"{pname} (general CSR SpMV + fused <p,Ap> tile partials), timed inside the CG loop "
                                              "of the same matrix with the coded form switched off",
                    "achieved": spmv_bytes / pus / 1e3, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": spmv_bytes / pus / 1e3 / HBM_PEAK_GBPS, "traffic": traffic_of("kernels", "spmv_plain", pname),
                    "avg_launch_us": pus, "launches_timed": pcnt, "rocprofv3_avg_us": rocprof_us(pname),
                    "frac_rocprofv3": None if rocprof_us(pname) is None else spmv_bytes / rocprof_us(pname) / 1e3 / HBM_PEAK_GBPS,
                    "algorithmic_bytes_per_launch": spmv_bytes, "bytes_are": "SURVEY 8d: nnz*12 + (n+1)*4 + 2n*8",
                    "why": "the north star's kernel (target >= 0.70); cg_iters_per_sec on these kernels: "
                           f"{pst2.iterations / pdt:.0f}"}
            finally:
                h.set_path(plain_only=False)
        # ---- HBM-resident leg: the same CG kernels on BASELINE config 5's matrix (8000 x 8000, N = 64 M) on this one device:
        # vectors of 512 MB each cannot sit in the 256 MiB Infinity Cache
        if nx == NX and not args.no_n64m and not strong:
            nb = 8000
            Ab = create_poisson_2d_csr(nb, nb, device=dev)
            bb = torch.ones(nb * nb, dtype=torch.float64, device=dev)
            hb = _hipk.handle_for(Ab)
            big = cg_legs(hb, bb, nb * nb, "kernels_n64m", "hbm")
            bdom = max(big, key=lambda k: k["avg_launch_us"])
            roof["legs"]["cg_dominant_n64m"] = roof_of(bdom, "longest kernel of the CG iteration at N = 64 M (BASELINE config 5's system on one "
                                                             "device): vectors of 512 MB, HBM-resident")
            roof["legs"]["cg_kernels_n64m"] = big
            del Ab, bb, hb
            _hipk.clear_cache()
            torch.cuda.empty_cache()

    # ---- launch-bound sizes: per-iteration time of whole cg / bicgstab solves through the C entry points on
    # systems the one-launch loops take (csrc/hipk_cg_mid.h, hipk_bi_mid.h); fixed iteration counts, b = ones, wall clock around
    # the call with the device drained on both sides.  Reported beside the headline, never part of `value`.
    launch_bound = None
    if not use_dist and nx == NX and not strong and not args.no_n64m:
        launch_bound = []
        from pytorch_sparse_solver.utils.matrix_utils import create_convdiff_2d_csr
        def grid3d(m_):   # 7-point Poisson on an m^3 grid
            import numpy as np
            import scipy.sparse as sp
            T_ = sp.diags([-1.0, 2.0, -1.0], [-1, 0, 1], shape=(m_, m_))
            I_ = sp.identity(m_)
            M_ = (sp.kron(sp.kron(T_, I_), I_) + sp.kron(sp.kron(I_, T_), I_) + sp.kron(sp.kron(I_, I_), T_)).tocsr()
            M_.sort_indices()
            return torch.sparse_csr_tensor(torch.from_numpy(M_.indptr.astype(np.int64)), torch.from_numpy(M_.indices.astype(np.int64)),
                                           torch.from_numpy(M_.data), size=M_.shape).to(dev)

        for solver, mk, side, its, kw in (("cg", create_poisson_2d_csr, 500, 1000, {}), ("cg", create_poisson_2d_csr, 1000, 1000, {}),
                                          ("cg", None, 64, 150, {}), ("bicgstab", create_convdiff_2d_csr, 500, 60, {}),
                                          ("gmres", create_convdiff_2d_csr, 500, 10, {"restart": 30})):
            Am = mk(side, side, device=dev) if mk is not None else grid3d(side)
            hm = _hipk.handle_for(Am)
            nm = Am.shape[0]
            bm = torch.ones(nm, dtype=torch.float64, device=dev)
            best = None
            for rep in range(3):
                xm = torch.zeros_like(bm)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                stm = _hipk.solve(solver, hm, bm, xm, tol=1e-12, atol=0.0, maxiter=its, **kw)
                torch.cuda.synchronize()
                us = (time.perf_counter() - t0) / max(stm.iterations, 1) * 1e6
                best = us if best is None or us < best else best
            launch_bound.append({"solver": solver + ("(30)" if solver == "gmres" else ""), "system": f"{side}^2 stencil" if mk is not None else f"{side}^3 7-point",
                                 "rows": nm, "reduction_chunks": -(-nm // 2048), "iterations_or_cycles": stm.iterations,
                                 ("us_per_cycle" if solver == "gmres" else "us_per_iteration"): round(best, 2)})
            del Am, hm, bm, xm
        _hipk.clear_cache()
        torch.cuda.empty_cache()

    if use_dist:
        # roofline leg at N > 1: this rank's local SpMV (no communication), HIP events on the launch stream,
        # on the bytes the selected path streams
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
        n_global = nx * nx if (strong or not use_dist) else nx * nx * world
        if strong:
            value = iters_total / dt
            unit = f"it/s (CG iterations of the ONE N={n_global} system, all ranks together)"
        else:
            value = world * iters_total / dt
            unit = "it/s (4M-row 5-pt Poisson CG iterations, summed over ranks)"
        out = {
            "metric": "cg_iters_per_sec",
            "value": value,
            "unit": unit,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload, "rows_global": n_global, "rows_per_gpu": n_rows_rank, "nnz_per_gpu": nnz_rank,
                       "iterations_per_solve": st.iterations, "info": info,
                       "relres": st.residual_norm / st.b_norm,
                       "step": "one full cg() solve via the public API" + (" (RowBlockCSR operand: this rank's rows)" if use_dist else ""),
                       "handle_creation_ms_outside_timed_region": handle_ms,
                       "cold_first_solve_ms": cold_first_solve_ms,
                       "placement_probe_GBps": getattr(st, "placement_GBps", None) or None,
                       "placement_allocations_drawn": getattr(st, "placement_tries", None) or None,
                       "rccl_ranks": world if use_dist else None,
                       "collectives": getattr(prob, "comm_kind", None) if use_dist else None},
            "spmv_standalone": spmv_standalone,
            "spmv": spmv_report,
            "kernels": kernels,
            "roofline": roof,
        }
        if launch_bound is not None:
            out["launch_bound_sizes"] = launch_bound
        if not args.no_cpu_baseline and not use_dist and nx == NX:
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
