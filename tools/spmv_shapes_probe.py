#!/usr/bin/env python3
"""SpMV GB/s on other sparsity structures (general path of the kernel): 7-pt / 27-pt 3-D stencils, random rows,
dense-as-CSR (BASELINE config 1 shape), power-law row lengths.  Dev aid."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd")]
import numpy as np, scipy.sparse as sp, torch
from pytorch_sparse_solver import _hipk
dev = "cuda:0"
def stencil3d(n, offs):
    N = n**3; idx = np.arange(N); i, j, k = idx // (n*n), (idx // n) % n, idx % n
    rows, cols = [], []
    for di, dj, dk in offs:
        ok = (i+di >= 0) & (i+di < n) & (j+dj >= 0) & (j+dj < n) & (k+dk >= 0) & (k+dk < n)
        rows.append(idx[ok]); cols.append(((i+di)*n*n + (j+dj)*n + (k+dk))[ok])
    r, c = np.concatenate(rows), np.concatenate(cols)
    return sp.csr_matrix((np.random.default_rng(0).standard_normal(r.size), (r, c)), shape=(N, N))
def timeit(h, x, y, reps=50):
    for _ in range(5): _hipk.spmv(h, x, out=y)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): _hipk.spmv(h, x, out=y)
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps
def run(name, S):
    S.sort_indices()
    h = _hipk.CsrHandle(torch.from_numpy(S.indptr.astype(np.int64)).to(dev), torch.from_numpy(S.indices.astype(np.int64)).to(dev), torch.from_numpy(S.data).to(dev), S.shape)
    x = torch.randn(S.shape[1], dtype=torch.float64, device=dev); y = torch.empty(S.shape[0], dtype=torch.float64, device=dev)
    ms = timeit(h, x, y); B = h.spmv_bytes()
    ref = torch.from_numpy(S @ x.cpu().numpy()).to(dev)
    err = (y - ref).abs().max().item() / max(ref.abs().max().item(), 1e-300)
    print(f"{name:34s} n={S.shape[0]:9d} nnz/row={S.nnz/S.shape[0]:7.1f}  {ms*1e3:9.1f} us  {B/ms/1e6:8.1f} GB/s  relerr={err:.1e}", flush=True)
off7 = [(0,0,0),(1,0,0),(-1,0,0),(0,1,0),(0,-1,0),(0,0,1),(0,0,-1)]
off27 = [(a,b,c) for a in (-1,0,1) for b in (-1,0,1) for c in (-1,0,1)]
run("7-pt 3D stencil 160^3", stencil3d(160, off7))
run("27-pt 3D stencil 120^3", stencil3d(120, off27))
rng = np.random.default_rng(1)
n = 1_000_000
def rand_rows(n, k, seed):
    r = np.random.default_rng(seed)
    cols = r.integers(0, n, (n, k)); cols.sort(axis=1)
    crow = np.arange(0, n * k + 1, k, dtype=np.int64)
    S = sp.csr_matrix((r.standard_normal(n * k), cols.ravel(), crow), shape=(n, n)); S.sum_duplicates(); return S
run("random 8/row", rand_rows(n, 8, 2))
run("random 50/row", rand_rows(400_000, 50, 3))
lens = np.minimum((rng.pareto(1.5, 200_000) * 5 + 1).astype(np.int64), 5000)
crow = np.zeros(200_001, dtype=np.int64); crow[1:] = np.cumsum(lens)
cols = rng.integers(0, 200_000, crow[-1]); S = sp.csr_matrix((rng.standard_normal(crow[-1]), cols, crow), shape=(200_000, 200_000)); S.sum_duplicates()
run("power-law rows (max 5000)", S)
D = sp.csr_matrix(rng.standard_normal((4000, 4000)))
run("dense 4000x4000 as CSR", D)
D1 = sp.csr_matrix(rng.standard_normal((1000, 1000)))
run("dense 1000x1000 as CSR (config 1)", D1)
