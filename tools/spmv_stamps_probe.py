#!/usr/bin/env python3
"""Where do the ~16 us of hipk_spmv_sell_wide_kernel<5,1,0> go?  
Run with HIPK_LIB_PATH=.../libhipk_stamps.so (make stamps): every wavefront of the kernel stamps the constant 100 MHz clock at
its phase boundaries.  One launch of the CG loop's SpMV form (y = A p with <p, y>) on the N = 4 M Poisson matrix after warm-ups;
prints, in microseconds: the spread of the wavefronts' START times (dispatch ramp), the per-phase durations (median / p90 over the
wavefronts) and the spread of their END times.  10 ns resolution."""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd"))
import numpy as np
import torch
from pytorch_sparse_solver import _hipk
from pytorch_sparse_solver.utils.matrix_utils import create_poisson_2d_csr
L = _hipk.lib()
if not hasattr(L, "hipk_debug_wide_stamps"):
    raise SystemExit("needs the stamps twin: HIPK_LIB_PATH=.../_lib/libhipk_stamps.so (make -C csrc stamps)")
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
dev = torch.device("cuda", 0)
A = create_poisson_2d_csr(nx, nx, device=dev)
h = _hipk.CsrHandle(A.crow_indices(), A.col_indices(), A.values(), A.shape)
n = nx * nx
p = torch.randn(n, dtype=torch.float64, device=dev, generator=torch.Generator(device=dev).manual_seed(0))
y = torch.empty_like(p)
part = torch.zeros(4096, dtype=torch.float64, device=dev)
s = torch.cuda.current_stream().cuda_stream


def launch():
    _hipk._check(L.hipk_spmv_ex(h.ptr, p.data_ptr(), y.data_ptr(), 1, p.data_ptr(), None, part.data_ptr(), None, None, 0, s), "spmv_ex")


for _ in range(50):
    launch()
torch.cuda.synchronize()
NS = 8
blocks = min(2048, int(L.hipk_chunk_count(n)))
results = []
for rep in range(5):
    L.hipk_debug_wide_stamps_clear()
    launch()
    buf = (ctypes.c_ulonglong * (2048 * 4 * NS))()
    L.hipk_debug_wide_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    _hipk._check(L.hipk_debug_wide_stamps(buf, len(buf)), "stamps")
    st = np.frombuffer(buf, dtype=np.uint64).reshape(2048, 4, NS).astype(np.int64)
    st = st[(st[:, :, 0] != 0).all(axis=1)]          # workgroups that ran (the launch's grid is padded to a multiple of 8)
    t0 = st[:, :, 0].min()
    us = (st - t0) / 100.0                            # 100 MHz -> us
    ph = {"load ucode + dictionary (first round trip)": us[:, :, 1] - us[:, :, 0], "barrier": us[:, :, 2] - us[:, :, 1],
          "per-lane tiles (grid-line ends)": us[:, :, 3] - us[:, :, 2], "first uniform tile": us[:, :, 4] - us[:, :, 3],
          "remaining uniform tiles": us[:, :, 5] - us[:, :, 4], "barrier before the fold": us[:, :, 6] - us[:, :, 5],
          "fold of the chunk partial": us[:, :, 7] - us[:, :, 6], "whole wavefront": us[:, :, 7] - us[:, :, 0]}
    wg_start = us[:, :, 0].min(axis=1)
    wg_end = us[:, :, 7].max(axis=1)
    late = wg_start > 2.0
    valid = (st[:, :, 4] != 0) & (st[:, :, 5] != 0)       # wavefronts that had a uniform tile of their parity
    ph["remaining uniform tiles"] = (us[:, :, 5] - us[:, :, 4])[valid]
    ph["first uniform tile"] = (us[:, :, 4] - us[:, :, 3])[valid]
    out = {"launch": rep, "workgroups": int(st.shape[0]), "kernel_span_us": float(us[:, :, 7].max()),
           "workgroups_starting_after_2us": int(late.sum()), "their_start_us_min_median_max": [float(np.min(wg_start[late])), float(np.median(wg_start[late])), float(np.max(wg_start[late]))] if late.any() else None,
           "end_of_the_on_time_workgroups_p50_p99_max_us": [float(np.percentile(wg_end[~late], q)) for q in (50, 99, 100)],
           "workgroup_duration_on_time_vs_late_median_us": [float(np.median((wg_end - wg_start)[~late])), float(np.median((wg_end - wg_start)[late])) if late.any() else None],
           "late_block_indices_sample": [int(i) for i in np.nonzero(late)[0][:24]],
           "start_p50_p90_max_us": [float(np.percentile(us[:, :, 0], q)) for q in (50, 90, 100)],
           "end_p10_p50_p90_us": [float(np.percentile(us[:, :, 7], q)) for q in (10, 50, 90)],
           "phases_median_p90_us": {k: [round(float(np.median(v)), 2), round(float(np.percentile(v, 90)), 2)] for k, v in ph.items()}}
    results.append(out)
    print(json.dumps(out), flush=True)
