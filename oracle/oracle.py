"""ctypes front end of the CPU oracle (oracle/krylov_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Never imported by the product package.

Every function takes/returns numpy arrays; CSR indices are int32, values fp64.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libkrylov_oracle.so")
_lib = None


class _Stats(ctypes.Structure):
    _fields_ = [
        ("iterations", ctypes.c_int64),
        ("matvecs", ctypes.c_int64),
        ("info", ctypes.c_int32),
        ("breakdown", ctypes.c_int32),
        ("b_norm", ctypes.c_double),
        ("residual_norm", ctypes.c_double),
        ("x_norm", ctypes.c_double),
        ("threshold", ctypes.c_double),
        ("recurrence_rs", ctypes.c_double),
    ]


@dataclass
class OracleResult:
    x: np.ndarray
    info: int
    iterations: int
    matvecs: int
    breakdown: int
    b_norm: float
    residual_norm: float
    x_norm: float
    threshold: float
    recurrence_rs: float


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (oracle/Makefile)."""
    if force or not os.path.exists(_LIB_PATH) or (
        os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "krylov_oracle.c"))
    ):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        dp = ctypes.POINTER(ctypes.c_double)
        ip = ctypes.POINTER(ctypes.c_int32)
        i64 = ctypes.c_int64
        L.orc_set_threads.argtypes = [ctypes.c_int]
        L.orc_get_threads.restype = ctypes.c_int
        L.orc_chunk_geom.argtypes = [i64, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
        L.orc_dot.argtypes = [i64, dp, dp]
        L.orc_dot.restype = ctypes.c_double
        L.orc_dot_parts.argtypes = [i64, dp, dp, dp]
        L.orc_dot_parts_ch.argtypes = [i64, ctypes.c_int, dp, dp, dp]
        L.orc_dot_tiled_parts_ch.argtypes = [i64, ctypes.c_int, dp, dp, dp]
        L.orc_dot_tiled.argtypes = [i64, dp, dp]
        L.orc_dot_tiled.restype = ctypes.c_double
        L.orc_reduce_parts.argtypes = [dp, ctypes.c_int]
        L.orc_reduce_parts.restype = ctypes.c_double
        L.orc_spmv.argtypes = [i64, ip, ip, dp, dp, dp, dp]
        sp = ctypes.POINTER(_Stats)
        L.orc_cg.argtypes = [i64, ip, ip, dp, dp, dp, ctypes.c_double, ctypes.c_double, i64, sp]
        L.orc_bicgstab.argtypes = [i64, ip, ip, dp, dp, dp, ctypes.c_double, ctypes.c_double, i64, sp]
        L.orc_pcg_jacobi.argtypes = [i64, ip, ip, dp, dp, dp, dp, ctypes.c_double, ctypes.c_double, i64, sp]
        L.orc_bicgstab_jacobi.argtypes = [i64, ip, ip, dp, dp, dp, dp, ctypes.c_double, ctypes.c_double, i64, sp]
        L.orc_block_jacobi_apply.argtypes = [i64, ctypes.c_int, dp, dp, dp]
        L.orc_pcg_blockjacobi.argtypes = [i64, ip, ip, dp, ctypes.c_int, dp, dp, dp, ctypes.c_double, ctypes.c_double, i64, sp]
        L.orc_gmres.argtypes = [i64, ip, ip, dp, dp, dp, ctypes.c_double, ctypes.c_double, ctypes.c_int, i64,
                                ctypes.c_int, ctypes.c_int, sp]
        L.orc_gmres_jacobi.argtypes = [i64, ip, ip, dp, dp, dp, dp, ctypes.c_double, ctypes.c_double, ctypes.c_int, i64,
                                       ctypes.c_int, ctypes.c_int, sp]
        # fp32-storage variant (same source compiled with -DORC_F32): vectors/values float, dots and scalars fp64
        fp = ctypes.POINTER(ctypes.c_float)
        L.orc32_dot.argtypes = [i64, fp, fp]
        L.orc32_dot.restype = ctypes.c_double
        L.orc32_dot_tiled.argtypes = [i64, fp, fp]
        L.orc32_dot_tiled.restype = ctypes.c_double
        L.orc32_spmv.argtypes = [i64, ip, ip, fp, fp, fp, fp]
        L.orc32_cg.argtypes = [i64, ip, ip, fp, fp, fp, ctypes.c_double, ctypes.c_double, i64, sp]
        L.orc32_bicgstab.argtypes = [i64, ip, ip, fp, fp, fp, ctypes.c_double, ctypes.c_double, i64, sp]
        L.orc32_pcg_jacobi.argtypes = [i64, ip, ip, fp, fp, fp, fp, ctypes.c_double, ctypes.c_double, i64, sp]
        L.orc32_gmres.argtypes = [i64, ip, ip, fp, fp, fp, ctypes.c_double, ctypes.c_double, ctypes.c_int, i64,
                                  ctypes.c_int, ctypes.c_int, sp]
        L.orc32_gmres_jacobi.argtypes = [i64, ip, ip, fp, fp, fp, fp, ctypes.c_double, ctypes.c_double, ctypes.c_int, i64,
                                         ctypes.c_int, ctypes.c_int, sp]
        L.orc32_set_threads.argtypes = [ctypes.c_int]
        _lib = L
    return _lib


def set_threads(t: int) -> None:
    lib().orc_set_threads(int(t))
    lib().orc32_set_threads(int(t))


def _d(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def _i(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))


def _csr(crow, col, val):
    crow = np.ascontiguousarray(crow, dtype=np.int32)
    col = np.ascontiguousarray(col, dtype=np.int32)
    val = np.ascontiguousarray(val, dtype=np.float64)
    return crow, col, val


def chunk_geom(n: int):
    ch, g = ctypes.c_int(), ctypes.c_int()
    lib().orc_chunk_geom(int(n), ctypes.byref(ch), ctypes.byref(g))
    return ch.value, g.value


def dot(a, b) -> float:
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    return float(lib().orc_dot(a.size, _d(a), _d(b)))


def dot_parts(a, b) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    _, g = chunk_geom(a.size)
    parts = np.zeros(g, dtype=np.float64)
    lib().orc_dot_parts(a.size, _d(a), _d(b), _d(parts))
    return parts


def dot_parts_ch(a, b, ch: int) -> np.ndarray:
    """Chunk partials of <a,b> with an explicit chunk size (a row block of a partitioned vector)."""
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    g = (a.size + ch - 1) // ch
    parts = np.zeros(max(g, 0), dtype=np.float64)
    if a.size:
        lib().orc_dot_parts_ch(a.size, int(ch), _d(a), _d(b), _d(parts))
    return parts


def dot_tiled(a, b) -> float:
    """The dot an SpMV epilogue produces ("tiled dot" spec, krylov_oracle.c header)."""
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    return float(lib().orc_dot_tiled(a.size, _d(a), _d(b)))


def dot_tiled_parts_ch(a, b, ch: int) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    g = (a.size + ch - 1) // ch
    parts = np.zeros(max(g, 0), dtype=np.float64)
    if a.size:
        lib().orc_dot_tiled_parts_ch(a.size, int(ch), _d(a), _d(b), _d(parts))
    return parts


def reduce_parts(parts) -> float:
    parts = np.ascontiguousarray(parts, dtype=np.float64)
    return float(lib().orc_reduce_parts(_d(parts), parts.size))


def spmv(crow, col, val, x, bsub=None) -> np.ndarray:
    crow, col, val = _csr(crow, col, val)
    x = np.ascontiguousarray(x, dtype=np.float64)
    n = crow.size - 1
    y = np.empty(n, dtype=np.float64)
    if bsub is not None:
        bsub = np.ascontiguousarray(bsub, dtype=np.float64)
    lib().orc_spmv(n, _i(crow), _i(col), _d(val), _d(x), _d(bsub) if bsub is not None else None, _d(y))
    return y


def _result(x, st: _Stats) -> OracleResult:
    return OracleResult(x=x, info=int(st.info), iterations=int(st.iterations), matvecs=int(st.matvecs),
                        breakdown=int(st.breakdown), b_norm=st.b_norm, residual_norm=st.residual_norm,
                        x_norm=st.x_norm, threshold=st.threshold, recurrence_rs=st.recurrence_rs)


def _prep(crow, col, val, b, x0):
    crow, col, val = _csr(crow, col, val)
    b = np.ascontiguousarray(b, dtype=np.float64)
    x = np.zeros_like(b) if x0 is None else np.array(x0, dtype=np.float64, copy=True)
    return crow, col, val, b, x


def cg(crow, col, val, b, x0=None, tol=1e-5, atol=0.0, maxiter=None) -> OracleResult:
    crow, col, val, b, x = _prep(crow, col, val, b, x0)
    st = _Stats()
    lib().orc_cg(b.size, _i(crow), _i(col), _d(val), _d(b), _d(x), float(tol), float(atol),
                 -1 if maxiter is None else int(maxiter), ctypes.byref(st))
    return _result(x, st)


def pcg_jacobi(crow, col, val, dinv, b, x0=None, tol=1e-5, atol=0.0, maxiter=None) -> OracleResult:
    """CG with M = diag(dinv) (TSL:806-856 with a non-identity M): restates hipk_pcg_solve."""
    crow, col, val, b, x = _prep(crow, col, val, b, x0)
    dinv = np.ascontiguousarray(dinv, dtype=np.float64)
    st = _Stats()
    lib().orc_pcg_jacobi(b.size, _i(crow), _i(col), _d(val), _d(dinv), _d(b), _d(x), float(tol), float(atol),
                         -1 if maxiter is None else int(maxiter), ctypes.byref(st))
    return _result(x, st)


def block_jacobi_apply(binv, v) -> np.ndarray:
    """z = blockdiag(binv) v; binv: [nb, bs, bs] (restates hipk_block_jacobi_kernel)."""
    binv = np.ascontiguousarray(binv, dtype=np.float64)
    v = np.ascontiguousarray(v, dtype=np.float64)
    out = np.empty_like(v)
    lib().orc_block_jacobi_apply(v.size, int(binv.shape[1]), _d(binv), _d(v), _d(out))
    return out


def pcg_blockjacobi(crow, col, val, binv, b, x0=None, tol=1e-5, atol=0.0, maxiter=None) -> OracleResult:
    """CG with M = blockdiag(binv) applied as a callable between the fused kernels (TSL:806-856 with a non-identity M)."""
    crow, col, val, b, x = _prep(crow, col, val, b, x0)
    binv = np.ascontiguousarray(binv, dtype=np.float64)
    st = _Stats()
    lib().orc_pcg_blockjacobi(b.size, _i(crow), _i(col), _d(val), int(binv.shape[1]), _d(binv), _d(b), _d(x), float(tol),
                              float(atol), -1 if maxiter is None else int(maxiter), ctypes.byref(st))
    return _result(x, st)


def pcg_jacobi32(crow, col, val, dinv, b, x0=None, tol=1e-5, atol=0.0, maxiter=None) -> OracleResult:
    crow, col, val, b, x = _prep32(crow, col, val, b, x0)
    dinv = _f32(dinv)
    st = _Stats()
    lib().orc32_pcg_jacobi(b.size, _i(crow), _i(col), _f(val), _f(dinv), _f(b), _f(x), float(tol), float(atol),
                           -1 if maxiter is None else int(maxiter), ctypes.byref(st))
    return _result(x, st)


def bicgstab(crow, col, val, b, x0=None, tol=1e-5, atol=0.0, maxiter=None) -> OracleResult:
    crow, col, val, b, x = _prep(crow, col, val, b, x0)
    st = _Stats()
    lib().orc_bicgstab(b.size, _i(crow), _i(col), _d(val), _d(b), _d(x), float(tol), float(atol),
                       -1 if maxiter is None else int(maxiter), ctypes.byref(st))
    return _result(x, st)


def gmres(crow, col, val, b, x0=None, tol=1e-5, atol=0.0, restart=20, maxiter=None,
          solve_method="batched", gpu_tolerances=False) -> OracleResult:
    crow, col, val, b, x = _prep(crow, col, val, b, x0)
    st = _Stats()
    method = {"batched": 0, "incremental": 1}[solve_method]
    rc = lib().orc_gmres(b.size, _i(crow), _i(col), _d(val), _d(b), _d(x), float(tol), float(atol), int(restart),
                         -1 if maxiter is None else int(maxiter), method, 1 if gpu_tolerances else 0,
                         ctypes.byref(st))
    if rc != 0:
        raise ValueError("oracle gmres supports 1 <= restart <= 127")
    return _result(x, st)


def bicgstab_jacobi(crow, col, val, dinv, b, x0=None, tol=1e-5, atol=0.0, maxiter=None) -> OracleResult:
    """BiCGStab with M = diag(dinv) applied before A (TSL:908, 922): restates hipk_pbicgstab_solve."""
    crow, col, val, b, x = _prep(crow, col, val, b, x0)
    dinv = np.ascontiguousarray(dinv, dtype=np.float64)
    st = _Stats()
    lib().orc_bicgstab_jacobi(b.size, _i(crow), _i(col), _d(val), _d(dinv), _d(b), _d(x), float(tol), float(atol),
                              -1 if maxiter is None else int(maxiter), ctypes.byref(st))
    return _result(x, st)


def gmres_jacobi(crow, col, val, dinv, b, x0=None, tol=1e-5, atol=0.0, restart=20, maxiter=None,
                 solve_method="batched", gpu_tolerances=False) -> OracleResult:
    """GMRES with M = diag(dinv) (left preconditioning, TSL:351, 750, 766, 791): restates hipk_pgmres_solve."""
    crow, col, val, b, x = _prep(crow, col, val, b, x0)
    dinv = np.ascontiguousarray(dinv, dtype=np.float64)
    st = _Stats()
    method = {"batched": 0, "incremental": 1}[solve_method]
    rc = lib().orc_gmres_jacobi(b.size, _i(crow), _i(col), _d(val), _d(dinv), _d(b), _d(x), float(tol), float(atol),
                                int(restart), -1 if maxiter is None else int(maxiter), method,
                                1 if gpu_tolerances else 0, ctypes.byref(st))
    if rc != 0:
        raise ValueError("oracle gmres supports 1 <= restart <= 127")
    return _result(x, st)


# ---------------------------------------------------------------------------------------------- fp32 storage
def _f(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def dot32(a, b) -> float:
    a, b = _f32(a), _f32(b)
    return float(lib().orc32_dot(a.size, _f(a), _f(b)))


def dot_tiled32(a, b) -> float:
    a, b = _f32(a), _f32(b)
    return float(lib().orc32_dot_tiled(a.size, _f(a), _f(b)))


def spmv32(crow, col, val, x, bsub=None) -> np.ndarray:
    crow, col = np.ascontiguousarray(crow, dtype=np.int32), np.ascontiguousarray(col, dtype=np.int32)
    val, x = _f32(val), _f32(x)
    y = np.empty(crow.size - 1, dtype=np.float32)
    if bsub is not None:
        bsub = _f32(bsub)
    lib().orc32_spmv(y.size, _i(crow), _i(col), _f(val), _f(x), _f(bsub) if bsub is not None else None, _f(y))
    return y


def _prep32(crow, col, val, b, x0):
    crow, col = np.ascontiguousarray(crow, dtype=np.int32), np.ascontiguousarray(col, dtype=np.int32)
    val, b = _f32(val), _f32(b)
    x = np.zeros_like(b) if x0 is None else np.array(x0, dtype=np.float32, copy=True)
    return crow, col, val, b, x


def cg32(crow, col, val, b, x0=None, tol=1e-5, atol=0.0, maxiter=None) -> OracleResult:
    crow, col, val, b, x = _prep32(crow, col, val, b, x0)
    st = _Stats()
    lib().orc32_cg(b.size, _i(crow), _i(col), _f(val), _f(b), _f(x), float(tol), float(atol),
                   -1 if maxiter is None else int(maxiter), ctypes.byref(st))
    return _result(x, st)


def bicgstab32(crow, col, val, b, x0=None, tol=1e-5, atol=0.0, maxiter=None) -> OracleResult:
    crow, col, val, b, x = _prep32(crow, col, val, b, x0)
    st = _Stats()
    lib().orc32_bicgstab(b.size, _i(crow), _i(col), _f(val), _f(b), _f(x), float(tol), float(atol),
                         -1 if maxiter is None else int(maxiter), ctypes.byref(st))
    return _result(x, st)


def gmres32(crow, col, val, b, x0=None, tol=1e-5, atol=0.0, restart=20, maxiter=None, solve_method="batched",
            gpu_tolerances=False) -> OracleResult:
    crow, col, val, b, x = _prep32(crow, col, val, b, x0)
    st = _Stats()
    method = {"batched": 0, "incremental": 1}[solve_method]
    rc = lib().orc32_gmres(b.size, _i(crow), _i(col), _f(val), _f(b), _f(x), float(tol), float(atol), int(restart),
                           -1 if maxiter is None else int(maxiter), method, 1 if gpu_tolerances else 0,
                           ctypes.byref(st))
    if rc != 0:
        raise ValueError("oracle gmres supports 1 <= restart <= 127")
    return _result(x, st)
