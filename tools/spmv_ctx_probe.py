#!/usr/bin/env python3
"""Where does the in-CG SpMV penalty come from?  Times the SpMV launch (a) plain, (b) with the fused dot, (c) with the
fused dot and 128 MB / 192 MB of unrelated vector streaming between launches (what cg_update/cg_direction do)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd")]
import torch
from pytorch_sparse_solver import _hipk
from pytorch_sparse_solver.utils.matrix_utils import create_poisson_2d_csr
dev = "cuda:0"; nx = 2000; n = nx * nx
A = create_poisson_2d_csr(nx, nx, device=dev); h = _hipk.handle_for(A); L = _hipk.lib()
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(n, dtype=torch.float64, device=dev, generator=g); y = torch.empty_like(x)
p0 = torch.zeros(2048, dtype=torch.float64, device=dev); p1 = torch.zeros(2048, dtype=torch.float64, device=dev)
vs = [torch.randn(n, dtype=torch.float64, device=dev, generator=g) for _ in range(4)]
s = torch.cuda.current_stream().cuda_stream
def spmv(mode): _hipk._check(L.hipk_spmv_ex(h.ptr, x.data_ptr(), y.data_ptr(), mode, x.data_ptr(), None, p0.data_ptr(), p1.data_ptr(), None, 0, s), "spmv_ex")
def timed(fn_between, mode, reps=100):
    evs = []
    for _ in range(10): spmv(mode); fn_between()
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); spmv(mode); b.record(); evs.append((a, b)); fn_between()
    torch.cuda.synchronize()
    return sum(a.elapsed_time(b) for a, b in evs) / reps * 1e3
def none(): pass
def stream128(): _hipk.axpy(0.5, vs[0], vs[1]); _hipk.axpy(0.5, vs[2], vs[3])   # 2 x (2 reads + 1 write) = 192 MB
def stream_xy(): _hipk.axpy(0.5, y, vs[1]); _hipk.xpby(vs[1], 0.5, x)             # touches y and rewrites x, like update/direction
print(f"plain spmv, back to back           : {timed(none, 0):6.1f} us")
print(f"spmv + fused dot (+combine)        : {timed(none, 1):6.1f} us")
print(f"spmv + dot, 192 MB unrelated stream: {timed(stream128, 1):6.1f} us")
print(f"spmv + dot, y read + x rewritten   : {timed(stream_xy, 1):6.1f} us")
print(f"plain spmv, x rewritten between    : {timed(stream_xy, 0):6.1f} us")
