#!/usr/bin/env python3
"""Quick on-GPU timing probe (development aid, not the bench): SpMV GB/s, dot/axpy GB/s, CG it/s."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd"))
import torch
from pytorch_sparse_solver import _hipk
from pytorch_sparse_solver.module_a import cg, get_last_stats
from pytorch_sparse_solver.utils.matrix_utils import create_poisson_2d_csr

def timeit(fn, reps, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

def main():
    nx = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    dev = "cuda:0"
    print(torch.cuda.get_device_name(0), "gfx950 devices:", _hipk.lib().hipk_device_count())
    A = create_poisson_2d_csr(nx, nx, device=dev)
    n = nx * nx
    h = _hipk.handle_for(A)
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(n, dtype=torch.float64, device=dev, generator=g)
    y = torch.empty_like(x)
    ms = timeit(lambda: _hipk.spmv(h, x, out=y), 200, 20)
    B = h.spmv_bytes()
    print(f"spmv  n={n} nnz={h.nnz}: {ms*1e3:.1f} us  {B/ms/1e6:.1f} GB/s ({B/ms/1e6/8000*100:.1f}% of 8 TB/s)")
    ref = torch.matmul(A, x)
    if not torch.allclose(y, ref, rtol=1e-12, atol=1e-12):
        raise SystemExit("gpu_probe: hipk SpMV differs from torch CSR matmul")
    ms_t = timeit(lambda: torch.matmul(A, x), 50, 5)
    print(f"torch CSR matmul: {ms_t*1e3:.1f} us  ({B/ms_t/1e6:.1f} GB/s on algorithmic bytes)")
    ms = timeit(lambda: _hipk.dot(x, y), 200, 20)
    print(f"dot   : {ms*1e3:.1f} us  {16*n/ms/1e6:.1f} GB/s")
    ms = timeit(lambda: _hipk.axpy(0.5, x, y), 200, 20)
    print(f"axpy  : {ms*1e3:.1f} us  {24*n/ms/1e6:.1f} GB/s")
    z = torch.empty_like(x)
    ms = timeit(lambda: z.copy_(x), 200, 20)
    print(f"torch copy: {ms*1e3:.1f} us  {16*n/ms/1e6:.1f} GB/s")
    b = torch.ones(n, dtype=torch.float64, device=dev)
    for rep in range(2):
        t0 = time.perf_counter()
        xs, info = cg(A, b, tol=1e-6)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        st = get_last_stats()
        if info != 0:
            raise SystemExit(f"gpu_probe: cg did not converge (info={info})")
        print(f"cg tol=1e-6: info={info} iters={st.iterations} wall={t1-t0:.3f}s dev={st.solve_ms:.1f}ms "
              f"-> {st.iterations/(st.solve_ms/1e3):.0f} it/s  relres={st.residual_norm/st.b_norm:.3e}")
    # profiled pass: SpMV kernel time inside CG
    bb = b.clone(); xx = torch.zeros_like(b)
    st = _hipk.solve("cg", h, bb, xx, tol=1e-6, atol=0.0, maxiter=300, profile=True)
    print(f"cg profiled: spmv_dot avg {st.spmv_ms_avg*1e3:.1f} us over {st.spmv_profiled} -> {B/st.spmv_ms_avg/1e6:.1f} GB/s; "
          f"iteration {st.solve_ms/300*1e3:.1f} us")

if __name__ == "__main__":
    # a probe must never report success after a failed solve / SpMV: any exception (HipkError, a HIP error surfacing at the
    # next synchronize) or a non-converged CG ends with a non-zero exit code; a GPU memory fault aborts the process (SIGABRT)
    try:
        main()
        torch.cuda.synchronize()
    except SystemExit:
        raise
    except BaseException as e:  # noqa: BLE001
        print(f"gpu_probe FAILED: {type(e).__name__}: {e}", file=sys.stderr)
        sys.exit(1)
