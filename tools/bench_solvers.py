#!/usr/bin/env python3
"""Secondary measurements (BASELINE configs 2-4) on one GPU: all three solvers through the public API.
Prints one JSON line per case; run on the GPU box:  python tools/bench_solvers.py > gpurun_out/solvers.jsonl"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd")]
import torch
from pytorch_sparse_solver import _hipk
from pytorch_sparse_solver.module_a import bicgstab, cg, get_last_stats, gmres
from pytorch_sparse_solver.utils.matrix_utils import create_convdiff_2d_csr, create_ldc_pressure_csr, create_poisson_2d_csr

DEV = "cuda:0"


def run(name, fn, A, b, reps=2, **kw):
    fn(A, b, **kw)                      # warm-up (handle creation, allocator)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        x, info = fn(A, b, **kw)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    st = get_last_stats()
    h = _hipk.handle_for(A)
    out = {"case": name, "n": h.n, "nnz": h.nnz, "info": info, "iterations_or_cycles": st.iterations,
           "matvecs": st.matvecs, "breakdown": st.breakdown, "relres": st.residual_norm / st.b_norm,
           "wall_ms": dt * 1e3, "device_ms": st.solve_ms, "matvecs_per_s": st.matvecs / dt,
           "iters_per_s": st.iterations / dt, "kwargs": {k: (v if isinstance(v, (int, float, str)) else type(v).__name__)
                                                         for k, v in kw.items()}}
    print(json.dumps(out), flush=True)


def main():
    nx = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    A = create_poisson_2d_csr(nx, nx, device=DEV)
    ones = torch.ones(nx * nx, dtype=torch.float64, device=DEV)
    run("config2_poisson_cg", cg, A, ones, tol=1e-6)
    run("config2_poisson_bicgstab", bicgstab, A, ones, tol=1e-6)
    run("config2_poisson_gmres30_batched", gmres, A, ones, reps=1, tol=1e-6, restart=30, maxiter=20)
    # Jacobi-preconditioned CG on a variable-coefficient diffusion matrix of the same size, fixed work
    from pytorch_sparse_solver.module_a import JacobiPreconditioner
    from pytorch_sparse_solver.utils.matrix_utils import create_variable_diffusion_2d_csr
    K = create_variable_diffusion_2d_csr(nx, nx, device=DEV)
    gk = torch.Generator(device=DEV).manual_seed(1)
    bk = torch.randn(nx * nx, dtype=torch.float64, device=DEV, generator=gk)
    run("vardiff_cg_plain_2000it", cg, K, bk, tol=1e-12, maxiter=2000)
    run("vardiff_cg_jacobi_2000it", cg, K, bk, tol=1e-12, maxiter=2000, M=JacobiPreconditioner(K))
    C = create_convdiff_2d_csr(nx, nx, device=DEV)
    g = torch.Generator(device=DEV).manual_seed(0)
    xt = torch.randn(nx * nx, dtype=torch.float64, device=DEV, generator=g)
    bc = _hipk.spmv(_hipk.handle_for(C), xt)
    run("config3_convdiff_bicgstab", bicgstab, C, bc, tol=1e-6)
    run("config3_convdiff_gmres30_batched", gmres, C, bc, reps=1, tol=1e-6, restart=30, maxiter=20)
    for lnx in (100, 1000):
        L = create_ldc_pressure_csr(lnx, device=DEV)
        gg = torch.Generator(device=DEV).manual_seed(3)
        bl = torch.randn(lnx * lnx, dtype=torch.float64, device=DEV, generator=gg)
        bl -= bl.mean()
        for m in ("batched", "incremental"):
            run(f"config4_ldc_nx{lnx}_gmres30_{m}", gmres, L, bl, tol=1e-10, maxiter=1000 if lnx == 100 else 20,
                restart=30, solve_method=m)
        run(f"config4_ldc_nx{lnx}_bicgstab", bicgstab, L, bl, tol=1e-10, maxiter=1000)
        # config 4 as BASELINE words it: fp32 storage (an extension here, the reference raises on an fp32 A)
        L32 = torch.sparse_csr_tensor(L.crow_indices(), L.col_indices(), L.values().float(), size=L.shape)
        run(f"config4_ldc_nx{lnx}_gmres30_batched_fp32", gmres, L32, bl.float(), tol=1e-5,
            maxiter=1000 if lnx == 100 else 20, restart=30, solve_method="batched")


if __name__ == "__main__":
    main()
