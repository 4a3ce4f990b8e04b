import os, sys, time, torch
sys.path[:0] = ["pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd"]
from pytorch_sparse_solver.module_a import cg, gmres, bicgstab, get_last_stats
from pytorch_sparse_solver.utils.matrix_utils import create_poisson_2d_csr, create_convdiff_2d_csr
dev = "cuda:0"
for nx in (128, 181, 256, 362, 512):
    A = create_convdiff_2d_csr(nx, nx, device=dev); P = create_poisson_2d_csr(nx, nx, device=dev)
    n = A.shape[0]
    b = torch.ones(n, dtype=torch.float64, device=dev)
    for name, fn, M, kw in (("gmres", gmres, A, dict(tol=1e-12, restart=30, maxiter=10)), ("cg", cg, P, dict(tol=1e-12, maxiter=300)), ("bicgstab", bicgstab, A, dict(tol=1e-14, maxiter=150))):
        for _ in range(2): fn(M, b, **kw)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        fn(M, b, **kw)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        st = get_last_stats()
        print(f"n={n:7d} {name:8s} {dt*1e3:8.3f} ms  units {st.iterations}  per unit {dt*1e6/max(st.iterations,1):8.2f} us", flush=True)
