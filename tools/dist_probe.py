#!/usr/bin/env python3
"""World-size-1 RCCL run of the row-partitioned CG: how much host overhead does the Python-driven loop add
over the single-device C loop?  (dev aid)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd")]
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import torch, torch.distributed as dist
from pytorch_sparse_solver.distributed import DistPoissonProblem, dist_cg
from pytorch_sparse_solver.module_a import cg, get_last_stats
from pytorch_sparse_solver.utils.matrix_utils import create_poisson_2d_csr
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
force_ch = int(sys.argv[2]) if len(sys.argv) > 2 else 0    # e.g. 16384 = the chunk size of an 8-rank weak-scaling run
ny = int(sys.argv[3]) if len(sys.argv) > 3 else nx           # grid-line width (8000 with nx = 1000, force_ch = 32768: a rank of config 5 at 8 ranks)
maxiter = int(sys.argv[4]) if len(sys.argv) > 4 else None
prob = DistPoissonProblem(nx_per_rank=nx, ny=ny, rank=0, world=1, device=dev, force_ch=force_ch)
print(f"chunk size {prob.part.ch}, local chunks {prob.part.g_local}")
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    x, info, st = dist_cg(prob, tol=1e-6, **({"maxiter": maxiter} if maxiter else {}))
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"dist_cg world=1: info={info} iters={st.iterations} {dt*1e3:.1f} ms -> {st.iterations/dt:.0f} it/s ({dt/st.iterations*1e6:.1f} us/iter)")
if ny != nx:
    dist.destroy_process_group(); sys.exit(0)
A = create_poisson_2d_csr(nx, nx, device=dev); b = torch.ones(nx*nx, dtype=torch.float64, device=dev)
cg(A, b, tol=1e-6); torch.cuda.synchronize(); t0 = time.perf_counter(); xr, _ = cg(A, b, tol=1e-6); torch.cuda.synchronize(); dt = time.perf_counter()-t0
print(f"cg single-device: iters={get_last_stats().iterations} {dt*1e3:.1f} ms; equal={torch.equal(x, xr)} (bitwise equality is expected only without a forced chunk size)")
dist.destroy_process_group()
