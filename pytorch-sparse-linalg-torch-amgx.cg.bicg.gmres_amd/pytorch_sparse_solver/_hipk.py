"""ctypes binding of libhipk.so (include/hipk.h) -- the MI355X kernels behind Module A.

There is NO CPU fallback in here: every function raises if the HIP library is missing
or no gfx950 device is visible.  `import torch` must come before the library is loaded
(torch's bundled libamdhip64.so.7 is then the HIP runtime the library binds to).
"""
from __future__ import annotations

import collections
import ctypes
import os
import threading
from dataclasses import dataclass
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# HIPK_LIB_PATH: load another build of the library (diagnostic builds, e.g. `make STAMPS=1`); the default is the in-tree one
LIB_PATH = os.environ.get("HIPK_LIB_PATH") or os.path.join(_HERE, "_lib", "libhipk.so")
CSRC_DIR = os.path.normpath(os.path.join(_HERE, "..", "csrc"))

HIPK_F32, HIPK_F64 = 0, 1
GMRES_BATCHED, GMRES_INCREMENTAL = 0, 1

# every symbol include/hipk.h declares (tests/test_abi.py checks the export list)
SYMBOLS = [
    "hipk_version", "hipk_build_id", "hipk_op_create", "hipk_placement_probe", "hipk_last_error", "hipk_device_count",
    "hipk_csr_create", "hipk_csr_destroy", "hipk_csr_rows", "hipk_csr_nnz", "hipk_csr_spmv_bytes",
    "hipk_csr_spmv_path", "hipk_last_spmv_kernel", "hipk_csr_set_path", "hipk_csr_format_bytes",
    "hipk_csr_transpose_work_bytes", "hipk_csr_transpose",
    "hipk_chunk_size", "hipk_chunk_count", "hipk_scratch_bytes",
    "hipk_spmv", "hipk_spmv_dot", "hipk_dot", "hipk_axpy", "hipk_xpby", "hipk_block_jacobi_apply",
    "hipk_cg_work_bytes", "hipk_cg_solve", "hipk_pcg_work_bytes", "hipk_pcg_solve", "hipk_pgmres_solve", "hipk_pbicgstab_work_bytes", "hipk_pbicgstab_solve", "hipk_pbicgstab_solve_cb", "hipk_pgmres_solve_cb",
    "hipk_bicgstab_work_bytes", "hipk_bicgstab_solve",
    "hipk_gmres_work_bytes", "hipk_gmres_solve",
    # step API (row-partitioned multi-GPU CG)
    "hipk_csr_create_ex", "hipk_spmv_ex", "hipk_dot_parts", "hipk_reduce_parts", "hipk_gather",
    "hipk_cg_scal_bytes", "hipk_cg_start", "hipk_cg_update", "hipk_cg_direction", "hipk_cg_xupdate",
    # step API: CG with a callable preconditioner
    "hipk_cgm_start", "hipk_cgm_direction",
    # row-partitioned CG, the loop of one rank in C
    "hipk_dist_cg_work_bytes", "hipk_dist_cg_solve", "hipk_dist_bicgstab_work_bytes", "hipk_dist_bicgstab_solve",
    "hipk_dist_gmres_work_bytes", "hipk_dist_gmres_solve",
    # experimental mailbox exchange provider for that loop
    "hipk_p2p_create", "hipk_p2p_create2", "hipk_p2p_export", "hipk_p2p_connect", "hipk_p2p_destroy", "hipk_p2p_error",
    "hipk_p2p_group_start", "hipk_p2p_group_end", "hipk_p2p_all_gather",
]


class Params(ctypes.Structure):
    _fields_ = [
        ("tol", ctypes.c_double),
        ("atol", ctypes.c_double),
        ("maxiter", ctypes.c_int64),
        ("restart", ctypes.c_int32),
        ("gmres_method", ctypes.c_int32),
        ("check_every", ctypes.c_int32),
        ("gpu_tolerances", ctypes.c_int32),
        ("profile", ctypes.c_int32),
        ("reserved", ctypes.c_int32),
    ]


class Stats(ctypes.Structure):
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
        ("solve_ms", ctypes.c_double),
        ("spmv_ms_avg", ctypes.c_double),
        ("spmv_profiled", ctypes.c_int64),
        ("dispatch_span_ms_avg", ctypes.c_double),
    ]


@dataclass
class SolveStats:
    """Side channel for what the reference never returns."""
    method: str
    iterations: int
    matvecs: int
    info: int
    breakdown: int
    b_norm: float
    residual_norm: float
    x_norm: float
    threshold: float
    recurrence_rs: float
    solve_ms: float
    spmv_ms_avg: float
    spmv_profiled: int
    dispatch_span_ms_avg: float = 0.0
    placement_GBps: float = 0.0    # large systems: what the placement probe read on the allocation the solve ran in (0: not probed)
    placement_tries: int = 0       #   allocations drawn (1: the first one was at the fast level or probing is off)


class HipkError(RuntimeError):
    pass


# hipk_rccl / hipk_dist_plan (include/hipk.h)
COLL_GROUP_FN = ctypes.CFUNCTYPE(ctypes.c_int)
COLL_ALLGATHER_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int,
                                     ctypes.c_void_p, ctypes.c_void_p)
COLL_SENDRECV_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int,
                                    ctypes.c_void_p, ctypes.c_void_p)


class Rccl(ctypes.Structure):
    _fields_ = [("group_start", ctypes.c_void_p), ("group_end", ctypes.c_void_p), ("all_gather", ctypes.c_void_p),
                ("send", ctypes.c_void_p), ("recv", ctypes.c_void_p), ("comm", ctypes.c_void_p), ("fused", ctypes.c_void_p)]


class DistPlan(ctypes.Structure):
    _fields_ = [("rank", ctypes.c_int32), ("world", ctypes.c_int32), ("n_local", ctypes.c_int64), ("n_ext", ctypes.c_int64),
                ("n_global", ctypes.c_int64), ("chunk_rows", ctypes.c_int32), ("g_red", ctypes.c_int32),
                ("per", ctypes.c_int32), ("halo_mode", ctypes.c_int32), ("n_send", ctypes.c_int32),
                ("n_ghost", ctypes.c_int32), ("slab", ctypes.c_int32), ("reserved", ctypes.c_int32),
                ("send_idx_dev", ctypes.c_void_p), ("ghost_src_dev", ctypes.c_void_p),
                ("send_off_dev", ctypes.c_void_p), ("dest_off_dev", ctypes.c_void_p),
                ("send_counts", ctypes.POINTER(ctypes.c_int32)), ("recv_counts", ctypes.POINTER(ctypes.c_int32)),
                ("send_first", ctypes.POINTER(ctypes.c_int64))]


# hipk_precond_fn / hipk_op_fn (include/hipk.h): int f(void *user, const void *in_dev, void *out_dev)
PRECOND_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p)
OP_FN = PRECOND_FN


_lib = None


def build(verbose: bool = False) -> str:
    """Compile libhipk.so for gfx950 with hipcc (csrc/Makefile). Works without a GPU."""
    import subprocess
    cmd = ["make", "-C", CSRC_DIR, "-j4"]
    if not verbose:
        cmd.insert(1, "-s")
    subprocess.check_call(cmd)
    return LIB_PATH


def lib():
    """Load libhipk.so; raise loudly when it is missing (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipkError(
            f"libhipk.so not found at {LIB_PATH}: the MI355X HIP extension is required for CUDA/ROCm tensors "
            f"(build it with `python __graft_entry__.py` or `make -C {CSRC_DIR}`); there is no CPU fallback.")
    L = ctypes.CDLL(LIB_PATH)
    vp, i64, i32, dbl = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_double
    L.hipk_version.restype = i32
    L.hipk_last_error.restype = ctypes.c_char_p
    L.hipk_build_id.restype = ctypes.c_char_p
    L.hipk_last_spmv_kernel.restype = ctypes.c_char_p
    L.hipk_device_count.restype = i32
    L.hipk_csr_create.argtypes = [ctypes.POINTER(vp), i64, i64, i64, vp, vp, i32, vp, i32, vp]
    L.hipk_csr_destroy.argtypes = [vp]
    L.hipk_op_create.argtypes = [ctypes.POINTER(vp), i64, i32, OP_FN, vp, vp]
    L.hipk_placement_probe.argtypes = [i64, vp, vp, vp, i32, i32, ctypes.POINTER(ctypes.c_double), vp]
    for f in (L.hipk_csr_rows, L.hipk_csr_nnz, L.hipk_csr_spmv_bytes, L.hipk_csr_format_bytes):
        f.argtypes = [vp]
        f.restype = i64
    L.hipk_csr_transpose_work_bytes.argtypes = [vp]
    L.hipk_csr_transpose_work_bytes.restype = ctypes.c_size_t
    L.hipk_csr_transpose.argtypes = [vp, vp, vp, vp, vp, ctypes.c_size_t, vp]
    L.hipk_csr_spmv_path.argtypes = [vp]
    L.hipk_csr_set_path.argtypes = [vp, i32]
    L.hipk_chunk_size.argtypes = [i64]
    L.hipk_chunk_count.argtypes = [i64]
    L.hipk_scratch_bytes.restype = ctypes.c_size_t
    L.hipk_spmv.argtypes = [vp, vp, vp, vp]
    L.hipk_spmv_dot.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    L.hipk_dot.argtypes = [i64, vp, vp, i32, vp, vp, vp]
    L.hipk_axpy.argtypes = [i64, dbl, vp, vp, i32, vp]
    L.hipk_xpby.argtypes = [i64, vp, dbl, vp, i32, vp]
    L.hipk_block_jacobi_apply.argtypes = [i64, i32, vp, vp, vp, i32, vp]
    for name in ("cg", "bicgstab"):
        wb = getattr(L, f"hipk_{name}_work_bytes")
        wb.argtypes = [i64, i32]
        wb.restype = ctypes.c_size_t
        getattr(L, f"hipk_{name}_solve").argtypes = [vp, vp, vp, vp, ctypes.c_size_t, ctypes.POINTER(Params),
                                                     ctypes.POINTER(Stats), vp]
    L.hipk_gmres_work_bytes.argtypes = [i64, i32, i32]
    L.hipk_gmres_work_bytes.restype = ctypes.c_size_t
    L.hipk_gmres_solve.argtypes = [vp, vp, vp, vp, ctypes.c_size_t, ctypes.POINTER(Params), ctypes.POINTER(Stats), vp]
    L.hipk_pcg_work_bytes.argtypes = [i64, i32]
    L.hipk_pcg_work_bytes.restype = ctypes.c_size_t
    L.hipk_pcg_solve.argtypes = [vp, vp, vp, vp, vp, ctypes.c_size_t, ctypes.POINTER(Params), ctypes.POINTER(Stats), vp]
    L.hipk_pgmres_solve.argtypes = [vp, vp, vp, vp, vp, ctypes.c_size_t, ctypes.POINTER(Params), ctypes.POINTER(Stats), vp]
    L.hipk_pbicgstab_work_bytes.argtypes = [i64, i32]
    L.hipk_pbicgstab_work_bytes.restype = ctypes.c_size_t
    L.hipk_pbicgstab_solve.argtypes = [vp, vp, vp, vp, vp, ctypes.c_size_t, ctypes.POINTER(Params), ctypes.POINTER(Stats), vp]
    L.hipk_pbicgstab_solve_cb.argtypes = [vp, PRECOND_FN, vp, vp, vp, vp, ctypes.c_size_t, ctypes.POINTER(Params),
                                          ctypes.POINTER(Stats), vp]
    L.hipk_pgmres_solve_cb.argtypes = L.hipk_pbicgstab_solve_cb.argtypes
    dbl = ctypes.c_double
    L.hipk_csr_create_ex.argtypes = [ctypes.POINTER(vp), i64, i64, i64, vp, vp, i32, vp, i32, i32, vp]
    L.hipk_spmv_ex.argtypes = [vp, vp, vp, i32, vp, vp, vp, vp, vp, i64, vp]
    L.hipk_dot_parts.argtypes = [i64, i32, vp, vp, i32, vp, vp]
    L.hipk_reduce_parts.argtypes = [vp, i32, vp, vp]
    L.hipk_gather.argtypes = [i64, vp, vp, vp, i32, vp]
    L.hipk_cg_scal_bytes.restype = ctypes.c_size_t
    L.hipk_cg_start.argtypes = [i64, i32, i32, vp, vp, vp, vp, vp, i32, dbl, dbl, i64, vp]
    L.hipk_cg_update.argtypes = [i64, i32, i32, vp, i64, vp, vp, vp, vp, i32, vp]
    L.hipk_cg_direction.argtypes = [i64, i32, i32, vp, i64, i64, vp, vp, vp, vp, vp, i32, vp]
    L.hipk_cg_xupdate.argtypes = [i64, i32, i32, vp, i64, vp, vp, vp, i32, vp]
    L.hipk_cgm_start.argtypes = [i64, i32, i32, vp, vp, vp, vp, vp, vp, i32, dbl, dbl, i64, vp]
    L.hipk_cgm_direction.argtypes = [i64, i32, i32, vp, i64, i64, vp, vp, vp, vp, vp, vp, i32, vp]
    L.hipk_dist_cg_work_bytes.argtypes = [ctypes.POINTER(DistPlan)]
    L.hipk_dist_cg_work_bytes.restype = ctypes.c_size_t
    L.hipk_dist_cg_solve.argtypes = [vp, ctypes.POINTER(DistPlan), ctypes.POINTER(Rccl), vp, vp, vp, ctypes.c_size_t,
                                     ctypes.POINTER(Params), ctypes.POINTER(Stats), vp]
    L.hipk_dist_bicgstab_work_bytes.argtypes = [ctypes.POINTER(DistPlan)]
    L.hipk_dist_bicgstab_work_bytes.restype = ctypes.c_size_t
    L.hipk_dist_bicgstab_solve.argtypes = [vp, ctypes.POINTER(DistPlan), ctypes.POINTER(Rccl), vp, vp, vp, ctypes.c_size_t,
                                           ctypes.POINTER(Params), ctypes.POINTER(Stats), vp]
    L.hipk_dist_gmres_work_bytes.argtypes = [ctypes.POINTER(DistPlan), i32]
    L.hipk_dist_gmres_work_bytes.restype = ctypes.c_size_t
    L.hipk_dist_gmres_solve.argtypes = [vp, ctypes.POINTER(DistPlan), ctypes.POINTER(Rccl), vp, vp, vp, ctypes.c_size_t,
                                        ctypes.POINTER(Params), ctypes.POINTER(Stats), vp]
    L.hipk_p2p_create.argtypes = [ctypes.POINTER(vp), i32, i32, ctypes.c_size_t]
    L.hipk_p2p_create2.argtypes = [ctypes.POINTER(vp), i32, i32, ctypes.c_size_t, i32, i32]
    L.hipk_p2p_export.argtypes = [vp, ctypes.c_char_p]
    L.hipk_p2p_connect.argtypes = [vp, ctypes.c_char_p]
    L.hipk_p2p_destroy.argtypes = [vp]
    L.hipk_p2p_error.argtypes = [vp]
    L.hipk_p2p_all_gather.argtypes = [vp, vp, ctypes.c_size_t, i32, vp, vp]
    _lib = L
    return L


def _check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().hipk_last_error().decode("utf-8", "replace")
        raise HipkError(f"{what} failed (status {rc}): {msg}")


def _stream(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def _dtype_code(dt: torch.dtype) -> int:
    if dt == torch.float64:
        return HIPK_F64
    if dt == torch.float32:
        return HIPK_F32
    raise HipkError(f"unsupported dtype {dt}")


class _SolveLock:
    """threading.Lock that refuses re-entry from the thread that holds it: a preconditioner callable that solves
    with the SAME matrix object would otherwise dead-lock (and would share the handle's scratch with the outer solve)."""

    def __init__(self):
        self._lock = threading.Lock()
        self._owner = None

    def __enter__(self):
        me = threading.get_ident()
        if self._owner == me:
            raise HipkError("nested solve on the same matrix from inside a preconditioner: the handle's scratch is in use "
                            "by the outer solve -- give the inner solver its own copy of the matrix (A.clone())")
        self._lock.acquire()
        self._owner = me
        return self

    def __exit__(self, *exc):
        self._owner = None
        self._lock.release()
        return False


class CsrHandle:
    """Owns a hipk_csr_t and keeps the tensors it borrows alive."""

    def __init__(self, crow: torch.Tensor, col: torch.Tensor, val: torch.Tensor, shape):
        if not val.is_cuda:
            raise HipkError("CsrHandle needs CUDA/ROCm tensors")
        self.crow = crow.contiguous()
        self.col = col.contiguous()
        self.val = val.contiguous()
        self.shape = (int(shape[0]), int(shape[1]))
        self.device = val.device
        self.dtype = val.dtype
        if self.crow.dtype != self.col.dtype or self.crow.dtype not in (torch.int32, torch.int64):
            raise HipkError("CSR indices must both be int32 or both int64")
        if self.crow.numel() != self.shape[0] + 1 or self.col.numel() != self.val.numel():
            raise HipkError("inconsistent CSR component sizes")
        self._h = ctypes.c_void_p()
        # the handle owns scratch its solves share (tile sums of the fused dots, the pinned signal words): one solve
        # at a time per handle; ctypes drops the GIL during a solve, so threads are serialised here
        self._lock = _SolveLock()
        with torch.cuda.device(self.device):
            rc = lib().hipk_csr_create(ctypes.byref(self._h), self.shape[0], self.shape[1], self.val.numel(),
                                       self.crow.data_ptr(), self.col.data_ptr(), self.crow.element_size(),
                                       self.val.data_ptr(), _dtype_code(self.dtype), _stream(self.device))
        _check(rc, "hipk_csr_create")

    @property
    def ptr(self):
        return self._h

    @property
    def n(self) -> int:
        return self.shape[0]

    @property
    def nnz(self) -> int:
        return int(self.val.numel())

    def spmv_bytes(self) -> int:
        return int(lib().hipk_csr_spmv_bytes(self._h))

    PATHS = {0: "tile_fast", 1: "tile", 2: "rowwave", 3: "coded", 4: "offset_coded"}

    def path(self) -> str:
        """SpMV kernel family selected by the structure analysis (include/hipk.h, hipk_spmv_path)."""
        return self.PATHS[int(lib().hipk_csr_spmv_path(self._h))]

    @staticmethod
    def last_spmv_kernel() -> str:
        """Kernel instantiation this thread's most recent SpMV launch selected (hipk_last_spmv_kernel)."""
        return lib().hipk_last_spmv_kernel().decode()

    def set_path(self, plain_only: bool) -> None:
        """plain_only=True: never use the coded form (A/B measurements, parity tests)."""
        _check(lib().hipk_csr_set_path(self._h, 1 if plain_only else 0), "hipk_csr_set_path")

    def format_bytes(self) -> int:
        """Bytes one SpMV moves in the format the selected path streams."""
        return int(lib().hipk_csr_format_bytes(self._h))

    def device_bytes(self) -> int:
        """Device memory this handle keeps alive: the CSR component tensors, the library's int32 index copies and
        tile-sum scratch, and the coded planes (format_bytes minus the two vectors an SpMV also moves)."""
        n, nnz, sv = self.shape[0], self.nnz, self.val.element_size()
        b = (self.crow.numel() + self.col.numel()) * self.crow.element_size() + nnz * sv
        b += 4 * (n + 1) + 4 * nnz + 64 * ((n + 255) // 256 + 1)
        if self.path() in ("coded", "offset_coded"):
            b += max(0, self.format_bytes() - 2 * n * sv)
        return int(b)

    def transposed(self) -> "CsrHandle":
        """Handle of A^T, built on the device from this handle's int32 arrays (hipk_csr_transpose: stable sort of the entries
        by column, so the rows of A^T come out column-sorted) and cached here: the adjoint solve of every backward pass
        reuses it (`ImplicitAdjointFunction.backward`, TSL:1237-1248)."""
        ht = getattr(self, "_transposed", None)
        if ht is not None:
            return ht
        L = lib()
        n_rows, n_cols = self.shape
        with torch.cuda.device(self.device):
            crow_t = torch.empty(n_cols + 1, dtype=torch.int32, device=self.device)
            col_t = torch.empty(max(self.nnz, 1), dtype=torch.int32, device=self.device)[:self.nnz]
            val_t = torch.empty(max(self.nnz, 1), dtype=self.dtype, device=self.device)[:self.nnz]
            wb = int(L.hipk_csr_transpose_work_bytes(self._h))
            work = torch.empty(wb, dtype=torch.uint8, device=self.device)
            _check(L.hipk_csr_transpose(self._h, crow_t.data_ptr(), col_t.data_ptr(), val_t.data_ptr(), work.data_ptr(), wb,
                                        _stream(self.device)), "hipk_csr_transpose")
            ht = CsrHandle(crow_t, col_t, val_t, (n_cols, n_rows))
        del work
        self._transposed = ht
        return ht

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            try:
                lib().hipk_csr_destroy(self._h)
            finally:
                self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# -------------------------------------------------------------------- handle cache
# Repeated solves with the same matrix (the LDC stepper: one matrix, thousands of RHS)
# reuse the analysed handle.  Keyed on storage identity + version counters + the VIEW
# (strides, storage offset, conj/neg bits): a dense `A` and its view `A.T` share storage
# and version but are different matrices (the adjoint solve of the implicit-diff backward,
# TSL:1245, asks for exactly that pair).
# The cache pins the source tensors and the handles' own device arrays (int32 index copies,
# coded planes): it is bounded by entries AND by bytes (HIPK_CACHE_BYTES, default 8 GiB; the
# most recent handle is always kept).  `clear_cache()` drops everything.
_CACHE: "collections.OrderedDict[tuple, tuple]" = collections.OrderedDict()
_CACHE_MAX = 8
_CACHE_BYTES_DEFAULT = 8 << 30


def _cache_budget() -> int:
    try:
        return int(os.environ.get("HIPK_CACHE_BYTES", _CACHE_BYTES_DEFAULT))
    except ValueError:
        return _CACHE_BYTES_DEFAULT


def _part_key(p: torch.Tensor):
    return (p.data_ptr(), p._version, p.numel(), tuple(p.stride()), p.storage_offset(), p.is_conj(), p.is_neg())


def _cache_key(A: torch.Tensor):
    if A.layout == torch.sparse_csr:
        parts = (A.crow_indices(), A.col_indices(), A.values())
    elif A.layout == torch.sparse_coo:
        parts = (A._indices(), A._values())
    elif A.layout == torch.sparse_csc:          # e.g. the transpose view of a CSR matrix (adjoint solves)
        parts = (A.ccol_indices(), A.row_indices(), A.values())
    elif A.layout == torch.strided:
        parts = (A,)
    else:
        raise HipkError(f"unsupported tensor layout {A.layout}")
    return (str(A.layout), tuple(A.shape), str(A.device), A.dtype) + tuple(_part_key(p) for p in parts)


def _pinned_bytes(h: "CsrHandle", src: torch.Tensor) -> int:
    """Device bytes an entry keeps alive: the handle's arrays + (when it was converted) the source tensor."""
    b = h.device_bytes()
    if src.layout == torch.strided:
        b += src.numel() * src.element_size()
    elif src.layout == torch.sparse_coo:
        b += src._indices().numel() * 8 + src._values().numel() * src._values().element_size()
    return b


def _store(key, h: "CsrHandle", src: torch.Tensor) -> None:
    _CACHE[key] = (h, src, _pinned_bytes(h, src))  # keep the source alive so its pointers cannot be recycled
    budget = _cache_budget()
    while len(_CACHE) > 1 and (len(_CACHE) > _CACHE_MAX or sum(e[2] for e in _CACHE.values()) > budget):
        _CACHE.popitem(last=False)


def handle_for(A: torch.Tensor) -> CsrHandle:
    """CSR handle of a CUDA tensor in any layout (dense / COO / CSR / CSC), converted once and cached."""
    key = _cache_key(A)
    hit = _CACHE.get(key)
    if hit is not None:
        _CACHE.move_to_end(key)
        return hit[0]
    src = A.detach()
    # transposed views (what the adjoint solve of the implicit-diff backward passes, TSL:1245): the handle of the base
    # matrix, transposed on the device -- no torch CSC -> CSR re-conversion, no second dense -> CSR pass
    if src.layout == torch.sparse_csc and src.dim() == 2:
        base = torch.sparse_csr_tensor(src.ccol_indices(), src.row_indices(), src.values(),
                                       size=(src.shape[1], src.shape[0]), check_invariants=False)
        h = handle_for(base).transposed()
        _store(key, h, src)
        return h
    if (src.layout == torch.strided and src.dim() == 2 and not src.is_contiguous() and src.t().is_contiguous()
            and not src.is_conj() and not src.is_neg()):
        h = handle_for(src.t()).transposed()
        _store(key, h, src)
        return h
    if src.layout == torch.sparse_csr:
        csr = src
    elif src.layout == torch.sparse_coo:
        csr = src.coalesce().to_sparse_csr()
    elif src.layout == torch.strided:
        csr = src.resolve_conj().resolve_neg().to_sparse_csr()
    else:
        csr = src.to_sparse_csr()
    h = CsrHandle(csr.crow_indices(), csr.col_indices(), csr.values(), csr.shape)
    _store(key, h, src)
    return h


def clear_cache() -> None:
    """Drop every cached handle (and the source matrices / index copies / coded planes they pin on the GPU)."""
    _CACHE.clear()


def cache_info():
    """(entries, pinned device bytes) of the handle cache."""
    return len(_CACHE), sum(e[2] for e in _CACHE.values())


# -------------------------------------------------------------------- primitives
def scratch(device) -> torch.Tensor:
    return torch.empty(int(lib().hipk_scratch_bytes()), dtype=torch.uint8, device=device)


def spmv(h: CsrHandle, x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    assert x.is_contiguous() and x.dtype == h.dtype and x.numel() == h.shape[1]
    y = torch.empty(h.shape[0], dtype=h.dtype, device=h.device) if out is None else out
    with torch.cuda.device(h.device):
        _check(lib().hipk_spmv(h.ptr, x.data_ptr(), y.data_ptr(), _stream(h.device)), "hipk_spmv")
    return y


def spmv_dot(h: CsrHandle, x: torch.Tensor, w: torch.Tensor):
    """(A x, <w, A x>) with the dot fused into the SpMV epilogue."""
    y = torch.empty(h.shape[0], dtype=h.dtype, device=h.device)
    out = torch.empty(1, dtype=torch.float64, device=h.device)
    sc = scratch(h.device)
    with torch.cuda.device(h.device):
        _check(lib().hipk_spmv_dot(h.ptr, x.data_ptr(), y.data_ptr(), w.data_ptr(), out.data_ptr(), sc.data_ptr(),
                                   _stream(h.device)), "hipk_spmv_dot")
    return y, out


def dot(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    assert x.is_cuda and x.is_contiguous() and y.is_contiguous() and x.dtype == y.dtype and x.numel() == y.numel()
    out = torch.empty(1, dtype=torch.float64, device=x.device)
    sc = scratch(x.device)
    with torch.cuda.device(x.device):
        _check(lib().hipk_dot(x.numel(), x.data_ptr(), y.data_ptr(), _dtype_code(x.dtype), out.data_ptr(),
                              sc.data_ptr(), _stream(x.device)), "hipk_dot")
    return out


def axpy(a: float, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """y <- y + a*x in place."""
    with torch.cuda.device(x.device):
        _check(lib().hipk_axpy(x.numel(), float(a), x.data_ptr(), y.data_ptr(), _dtype_code(x.dtype),
                               _stream(x.device)), "hipk_axpy")
    return y


def xpby(x: torch.Tensor, b: float, y: torch.Tensor) -> torch.Tensor:
    """y <- x + b*y in place."""
    with torch.cuda.device(x.device):
        _check(lib().hipk_xpby(x.numel(), x.data_ptr(), float(b), y.data_ptr(), _dtype_code(x.dtype),
                               _stream(x.device)), "hipk_xpby")
    return y


def block_jacobi_apply(binv: torch.Tensor, block_size: int, v: torch.Tensor) -> torch.Tensor:
    """z = blockdiag(binv) v on the device (hipk_block_jacobi_apply); binv: [ceil(n / bs), bs, bs] contiguous."""
    assert v.is_cuda and v.is_contiguous() and binv.is_contiguous() and binv.dtype == v.dtype and binv.device == v.device
    out = torch.empty_like(v)
    with torch.cuda.device(v.device):
        _check(lib().hipk_block_jacobi_apply(v.numel(), int(block_size), binv.data_ptr(), v.data_ptr(), out.data_ptr(),
                                             _dtype_code(v.dtype), _stream(v.device)), "hipk_block_jacobi_apply")
    return out


# -------------------------------------------------------------------- whole solves
# Systems whose vectors live in HBM (a vector beyond the 256 MiB Infinity Cache): the CG vector kernels run at one of two discrete
# speeds depending on where r, p and x landed PHYSICALLY -- vectors in separate allocations were at the slow level every time,
# vectors in ONE allocation at the fast one in about half of the draws (profiles/r02_axpy_realloc_64m.txt).  So x joins the work
# buffer's allocation, a probe with the direction step's memory shape (hipk_placement_probe, ~0.5 ms per pass at N = 64 M) reads the
# level, and a slow draw is replaced by a fresh allocation (the earlier ones are held until the choice is made, so that the
# allocator has to map new pages), at most HIPK_PLACEMENT_TRIES (4) in all; the best one is kept.  HIPK_PLACEMENT_PROBE=0: off.
_PLACEMENT_MIN_VECTOR_BYTES = 256 << 20
_PLACEMENT_FAST_GBPS = 5400.0   # between the two levels (5.85 and 4.9 TB/s on the 40 n bytes of a pass)


def _placed_cg_work(h, x: torch.Tensor, work_bytes: int):
    """(buffer, work view, x view, probe GB/s, tries) for a large CG solve, or None when the system is small / probing is off."""
    item = x.element_size()
    n = x.numel()
    if n * item <= _PLACEMENT_MIN_VECTOR_BYTES or os.environ.get("HIPK_PLACEMENT_PROBE", "1") == "0" or hasattr(h, "regions"):
        return None
    L = lib()
    vec = (n * item + 255) // 256 * 256
    tries = max(1, int(os.environ.get("HIPK_PLACEMENT_TRIES", "4")))
    r_off = 256 + int(L.hipk_scratch_bytes())          # hipk_cg_solve's layout: header, partials, r, p, Ap (csrc/hipk_cg.hip)
    held, best = [], None
    us = ctypes.c_double()
    for t in range(tries):
        buf = torch.empty(work_bytes + vec, dtype=torch.uint8, device=x.device)
        held.append(buf)
        xin = buf[work_bytes:work_bytes + n * item].view(x.dtype)
        xin.copy_(x)
        _check(L.hipk_placement_probe(n, buf.data_ptr() + r_off, buf.data_ptr() + r_off + vec, xin.data_ptr(), _dtype_code(x.dtype),
                                      3, ctypes.byref(us), _stream(x.device)), "hipk_placement_probe")
        rate = 5 * n * item / us.value / 1e3
        if best is None or rate > best[3]:
            best = (buf, buf[:work_bytes], xin, rate, t + 1)
        if rate >= _PLACEMENT_FAST_GBPS:
            break
    return best[0], best[1], best[2], best[3], len(held)


def _solve(method: str, h: CsrHandle, b: torch.Tensor, x: torch.Tensor, prm: Params, work_bytes: int) -> SolveStats:
    L = lib()
    placed = None
    with torch.cuda.device(h.device):
        if method == "cg":
            placed = _placed_cg_work(h, x, work_bytes)
    if placed is not None:
        _buf, work, x_run, probe_gbps, probe_tries = placed
    else:
        work, x_run, probe_gbps, probe_tries = torch.empty(work_bytes, dtype=torch.uint8, device=h.device), x, 0.0, 0
    if hasattr(h, "regions"):   # OpHandle: the operator callback resolves raw pointers against these tensors
        h.regions.append(work)
    st = Stats()
    fn = getattr(L, f"hipk_{method}_solve")
    with h._lock, torch.cuda.device(h.device):
        rc = fn(h.ptr, b.data_ptr(), x_run.data_ptr(), work.data_ptr(), work_bytes, ctypes.byref(prm), ctypes.byref(st),
                _stream(h.device))
    _check(rc, f"hipk_{method}_solve")
    if x_run is not x:
        x.copy_(x_run)
    return SolveStats(placement_GBps=probe_gbps, placement_tries=probe_tries, method=method, iterations=st.iterations, matvecs=st.matvecs, info=st.info,
                      breakdown=st.breakdown, b_norm=st.b_norm, residual_norm=st.residual_norm, x_norm=st.x_norm,
                      threshold=st.threshold, recurrence_rs=st.recurrence_rs, solve_ms=st.solve_ms,
                      spmv_ms_avg=st.spmv_ms_avg, spmv_profiled=st.spmv_profiled,
                      dispatch_span_ms_avg=st.dispatch_span_ms_avg)


def solve(method: str, h: CsrHandle, b: torch.Tensor, x: torch.Tensor, *, tol: float, atol: float,
          maxiter: Optional[int], restart: int = 20, solve_method: str = "batched", check_every: int = 0,
          profile: bool = False) -> SolveStats:
    """Run hipk_{cg,bicgstab,gmres}_solve. `x` holds x0 on entry and the solution on return."""
    if h.shape[0] != h.shape[1]:
        raise ValueError(f"linear operator must be a square matrix, but has shape: {h.shape}")
    assert b.is_contiguous() and x.is_contiguous() and b.dtype == h.dtype and x.dtype == h.dtype
    assert b.numel() == h.n and x.numel() == h.n and b.device == h.device and x.device == h.device
    prm = Params()
    prm.tol, prm.atol = float(tol), float(atol)
    prm.maxiter = -1 if maxiter is None else int(maxiter)
    prm.restart = int(restart)
    prm.gmres_method = {"batched": GMRES_BATCHED, "incremental": GMRES_INCREMENTAL}[solve_method]
    prm.check_every = int(check_every)
    prm.gpu_tolerances = 1  # device.type == 'cuda' on ROCm as well (TSL:737)
    prm.profile = int(profile)   # True/1: SpMV launches; CG only: 2 update kernel, 3 direction kernel
    L = lib()
    code = _dtype_code(h.dtype)
    if method == "gmres":
        wb = L.hipk_gmres_work_bytes(h.n, int(restart), code)
    else:
        wb = getattr(L, f"hipk_{method}_work_bytes")(h.n, code)
    return _solve(method, h, b, x, prm, int(wb))


def solve_pcg(h: CsrHandle, dinv: torch.Tensor, b: torch.Tensor, x: torch.Tensor, *, tol: float, atol: float,
              maxiter: Optional[int], check_every: int = 0, method: str = "cg") -> SolveStats:
    """hipk_pcg_solve / hipk_pbicgstab_solve (method "cg" / "bicgstab") with M = diag(dinv).
    `x` holds x0 on entry and the solution on return."""
    if h.shape[0] != h.shape[1]:
        raise ValueError(f"linear operator must be a square matrix, but has shape: {h.shape}")
    for t in (dinv, b, x):
        assert t.is_contiguous() and t.dtype == h.dtype and t.numel() == h.n and t.device == h.device
    prm = Params()
    prm.tol, prm.atol = float(tol), float(atol)
    prm.maxiter = -1 if maxiter is None else int(maxiter)
    prm.check_every = int(check_every)
    prm.gpu_tolerances = 1
    L = lib()
    name = {"cg": "pcg", "bicgstab": "pbicgstab"}[method]
    wb = int(getattr(L, f"hipk_{name}_work_bytes")(h.n, _dtype_code(h.dtype)))
    work = torch.empty(wb, dtype=torch.uint8, device=h.device)
    if hasattr(h, "regions"):   # OpHandle: the operator callback resolves raw pointers against these tensors
        h.regions.append(work)
    st = Stats()
    with h._lock, torch.cuda.device(h.device):
        rc = getattr(L, f"hipk_{name}_solve")(h.ptr, dinv.data_ptr(), b.data_ptr(), x.data_ptr(), work.data_ptr(), wb,
                                              ctypes.byref(prm), ctypes.byref(st), _stream(h.device))
    _check(rc, f"hipk_{name}_solve")
    return SolveStats(method=f"{name}_jacobi", iterations=st.iterations, matvecs=st.matvecs, info=st.info,
                      breakdown=st.breakdown, b_norm=st.b_norm, residual_norm=st.residual_norm, x_norm=st.x_norm,
                      threshold=st.threshold, recurrence_rs=st.recurrence_rs, solve_ms=st.solve_ms,
                      spmv_ms_avg=st.spmv_ms_avg, spmv_profiled=st.spmv_profiled,
                      dispatch_span_ms_avg=st.dispatch_span_ms_avg)


def solve_pgmres(h: CsrHandle, dinv: torch.Tensor, b: torch.Tensor, x: torch.Tensor, *, tol: float, atol: float,
                 maxiter: Optional[int], restart: int = 20, solve_method: str = "batched") -> SolveStats:
    """hipk_pgmres_solve: GMRES with M = diag(dinv) applied after every A (left preconditioning)."""
    if h.shape[0] != h.shape[1]:
        raise ValueError(f"linear operator must be a square matrix, but has shape: {h.shape}")
    for t in (dinv, b, x):
        assert t.is_contiguous() and t.dtype == h.dtype and t.numel() == h.n and t.device == h.device
    prm = Params()
    prm.tol, prm.atol = float(tol), float(atol)
    prm.maxiter = -1 if maxiter is None else int(maxiter)
    prm.restart = int(restart)
    prm.gmres_method = {"batched": GMRES_BATCHED, "incremental": GMRES_INCREMENTAL}[solve_method]
    prm.gpu_tolerances = 1
    L = lib()
    wb = int(L.hipk_gmres_work_bytes(h.n, int(restart), _dtype_code(h.dtype)))
    work = torch.empty(wb, dtype=torch.uint8, device=h.device)
    if hasattr(h, "regions"):   # OpHandle: the operator callback resolves raw pointers against these tensors
        h.regions.append(work)
    st = Stats()
    with h._lock, torch.cuda.device(h.device):
        rc = L.hipk_pgmres_solve(h.ptr, dinv.data_ptr(), b.data_ptr(), x.data_ptr(), work.data_ptr(), wb,
                                 ctypes.byref(prm), ctypes.byref(st), _stream(h.device))
    _check(rc, "hipk_pgmres_solve")
    return SolveStats(method="pgmres_jacobi", iterations=st.iterations, matvecs=st.matvecs, info=st.info,
                      breakdown=st.breakdown, b_norm=st.b_norm, residual_norm=st.residual_norm, x_norm=st.x_norm,
                      threshold=st.threshold, recurrence_rs=st.recurrence_rs, solve_ms=st.solve_ms,
                      spmv_ms_avg=st.spmv_ms_avg, spmv_profiled=st.spmv_profiled,
                      dispatch_span_ms_avg=st.dispatch_span_ms_avg)


def solve_cg_callable(h: CsrHandle, M, b: torch.Tensor, x: torch.Tensor, *, tol: float, atol: float,
                      maxiter: Optional[int]) -> SolveStats:
    """CG with a CALLABLE preconditioner on the fused kernels (`_cg_solve` with M, TSL:806-856):
    `solve_cg_stepwise` with the handle's SpMV.  With M = (r -> dinv * r) the result equals hipk_pcg_solve's
    bit for bit."""
    if h.shape[0] != h.shape[1]:
        raise ValueError(f"linear operator must be a square matrix, but has shape: {h.shape}")
    return solve_cg_stepwise(h, None, M, b, x, tol=tol, atol=atol, maxiter=maxiter)


def solve_cg_stepwise(h: Optional[CsrHandle], A_fn, M, b: torch.Tensor, x: torch.Tensor, *, tol: float, atol: float,
                      maxiter: Optional[int]) -> SolveStats:
    """CG driven from the host through the step API, for operands the device-resident loop cannot call itself:
    a CALLABLE preconditioner `M` (h given) and/or a MATRIX-FREE operator `A_fn` (h None) -- `_cg_solve`, TSL:806-856.

    Per iteration: Ap = A p with <p,Ap> (fused into hipk_spmv_ex, or A_fn + hipk_dot_parts) | hipk_cg_update (r, <r,r>) |
    [z = M(r), <r,z>] | hipk_cg_direction / hipk_cgm_direction.  `A_fn` / `M` are the caller's own device code,
    enqueued on the current stream; scalars and the stop word live on the device, the host looks at the stop word
    every 8, 16, 32, 64, 64, ... iterations.  `x` holds x0 on entry and the solution on return."""
    assert (h is None) != (A_fn is None)
    assert b.is_contiguous() and x.is_contiguous() and b.is_cuda and x.dtype == b.dtype and x.shape == b.shape
    assert b.dtype in (torch.float64, torch.float32)
    if h is not None:
        assert b.dtype == h.dtype and b.numel() == h.n and b.device == h.device
    L = lib()
    n, dev, dt = b.numel(), b.device, _dtype_code(b.dtype)
    ch, G = int(L.hipk_chunk_size(n)), int(L.hipk_chunk_count(n))
    maxit = 10 * n if maxiter is None else int(maxiter)
    MODE_DOT_W, MODE_DOT_YY, MODE_RESID = 1, 2, 4

    def apply(fn, v, what):
        z = fn(v)
        if not isinstance(z, torch.Tensor) or z.shape != v.shape:
            raise ValueError(f"the {what} must map a vector to a vector of the same shape")
        return z.to(device=dev, dtype=b.dtype).contiguous()

    import contextlib
    with (h._lock if h is not None else contextlib.nullcontext()), torch.cuda.device(dev):
        s = _stream(dev)
        r, p, Ap = torch.empty_like(b), torch.empty_like(b), torch.empty_like(b)
        parts = torch.zeros(6 * 2048, dtype=torch.float64, device=dev)
        part_pAp, part_rr, part_rz, part_bb, spare, part_xx = (parts[i * 2048:(i + 1) * 2048] for i in range(6))
        scal = torch.zeros(int(L.hipk_cg_scal_bytes()) // 8, dtype=torch.float64, device=dev)
        stop_word = scal[6:7].view(torch.int64)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()

        def dot_parts(u, v, part):
            _check(L.hipk_dot_parts(n, ch, u.data_ptr(), v.data_ptr(), dt, part.data_ptr(), s), "hipk_dot_parts")

        def residual(out, part_yy):                                  # out = b - A x  (+ <out,out> partials)
            if h is not None:
                _check(L.hipk_spmv_ex(h.ptr, x.data_ptr(), out.data_ptr(), MODE_RESID | MODE_DOT_YY, None, b.data_ptr(),
                                      spare.data_ptr(), part_yy.data_ptr(), None, 0, s), "hipk_spmv_ex")
            else:
                torch.sub(b, apply(A_fn, x, "operator"), out=out)
                dot_parts(out, out, part_yy)

        # r0 = b - A x0, <r0,r0>; <b,b>; [z0 = M r0, <r0,z0>]; p0 = z0 (or r0)   (TSL:815-826)
        residual(r, part_rr)
        dot_parts(b, b, part_bb)
        if M is not None:
            z = apply(M, r, "preconditioner")
            dot_parts(r, z, part_rz)
            _check(L.hipk_cgm_start(n, ch, G, scal.data_ptr(), part_rz.data_ptr(), part_rr.data_ptr(), part_bb.data_ptr(),
                                    z.data_ptr(), p.data_ptr(), dt, float(tol), float(atol), maxit, s), "hipk_cgm_start")
        else:
            _check(L.hipk_cg_start(n, ch, G, scal.data_ptr(), part_rr.data_ptr(), part_bb.data_ptr(), r.data_ptr(),
                                   p.data_ptr(), dt, float(tol), float(atol), maxit, s), "hipk_cg_start")
        it, stop, batch = 0, 1 << 62, 8
        while it < maxit:
            stop = int(stop_word.item())                      # one synchronisation per batch
            if stop <= it:
                break
            end = min(maxit, it + batch)
            batch = min(64, 2 * batch)
            while it < end:
                if h is not None:
                    _check(L.hipk_spmv_ex(h.ptr, p.data_ptr(), Ap.data_ptr(), MODE_DOT_W, p.data_ptr(), None,
                                          part_pAp.data_ptr(), spare.data_ptr(), stop_word.data_ptr(), it, s),
                           "hipk_spmv_ex")
                else:
                    Ap = apply(A_fn, p, "operator")
                    dot_parts(p, Ap, part_pAp)
                _check(L.hipk_cg_update(n, ch, G, scal.data_ptr(), it, part_pAp.data_ptr(), Ap.data_ptr(), r.data_ptr(),
                                        part_rr.data_ptr(), dt, s), "hipk_cg_update")
                if M is not None:
                    z = apply(M, r, "preconditioner")          # past the stop this works on a converged r: harmless
                    dot_parts(r, z, part_rz)
                    _check(L.hipk_cgm_direction(n, ch, G, scal.data_ptr(), it, maxit, part_pAp.data_ptr(),
                                                part_rz.data_ptr(), part_rr.data_ptr(), z.data_ptr(), p.data_ptr(),
                                                x.data_ptr(), dt, s), "hipk_cgm_direction")
                else:
                    _check(L.hipk_cg_direction(n, ch, G, scal.data_ptr(), it, maxit, part_pAp.data_ptr(),
                                               part_rr.data_ptr(), r.data_ptr(), p.data_ptr(), x.data_ptr(), dt, s),
                           "hipk_cg_direction")
                it += 1
        stop = int(stop_word.item())
        iterations = min(stop, it)
        # recurrence <r,r> of the last completed iteration, before the partial slots are reused
        out = torch.empty(4, dtype=torch.float64, device=dev)
        _check(L.hipk_reduce_parts(part_rr.data_ptr(), G, out[3:4].data_ptr(), s), "hipk_reduce_parts")
        # TSL:1007-1014: ||M (b - A x)||, ||x||
        res = torch.empty_like(b)
        residual(res, part_rz)
        if M is not None:
            zr = apply(M, res, "preconditioner")
            dot_parts(zr, zr, part_rz)
        dot_parts(x, x, part_xx)
        for k, prt in enumerate((part_rz, part_xx, part_bb)):
            _check(L.hipk_reduce_parts(prt.data_ptr(), G, out[k:k + 1].data_ptr(), s), "hipk_reduce_parts")
        e1.record()
        res2, xx, bs, rs = (float(v) for v in out.cpu())
    b_norm, res_norm = max(bs, 0.0) ** 0.5, max(res2, 0.0) ** 0.5
    x_norm = max(xx, 0.0) ** 0.5 if xx == xx else float("nan")
    thr = max(float(torch.tensor(tol, dtype=torch.float32)) * b_norm, float(torch.tensor(atol, dtype=torch.float32)))
    info = -1 if (x_norm != x_norm or res_norm > thr) else 0
    method = ("cg_callable_M" if M is not None else "cg") if h is not None else \
             ("cg_matrix_free_callable_M" if M is not None else "cg_matrix_free")
    return SolveStats(method=method, iterations=iterations, matvecs=iterations + 2, info=info, breakdown=0,
                      b_norm=b_norm, residual_norm=res_norm, x_norm=x_norm, threshold=thr, recurrence_rs=rs,
                      solve_ms=e0.elapsed_time(e1), spmv_ms_avg=0.0, spmv_profiled=0)


class OpHandle:
    """hipk_op_create: a handle WITHOUT a matrix -- every product of a solve is the caller's `fn(v) -> A v` (device code on
    the current stream), followed by the library's epilogue kernel (residual form, fused dots).  It quacks like a CsrHandle
    for `solve` / `solve_pcg` / `_solve_with_callback` (ptr, n, shape, dtype, device, _lock).  The C loop hands raw device
    pointers to the callback; they are resolved against the tensors registered in `regions` (the solve's workspace, b, x)."""

    def __init__(self, fn, n: int, dtype: torch.dtype, device):
        self.fn, self.n, self.shape, self.dtype, self.device = fn, int(n), (int(n), int(n)), dtype, torch.device(device)
        self.regions = []
        self.errors = []
        self.calls = 0
        self._lock = _SolveLock()
        self._nbytes = self.n * torch.empty(0, dtype=dtype).element_size()
        self._cb = OP_FN(self._call)      # keep the thunk alive as long as the handle
        self._h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _check(lib().hipk_op_create(ctypes.byref(self._h), self.n, _dtype_code(dtype), self._cb, None, _stream(self.device)),
                   "hipk_op_create")

    @property
    def ptr(self):
        return self._h

    def _view(self, ptr):
        ptr = int(ptr)
        for t in self.regions:
            off = ptr - t.data_ptr()
            if 0 <= off and off + self._nbytes <= t.numel() * t.element_size():
                flat = t.view(torch.uint8) if t.dtype != torch.uint8 else t
                return flat[off:off + self._nbytes].view(self.dtype)
        raise HipkError("operator callback: pointer outside the solve's workspace, b and x")

    def _call(self, _user, x_ptr, y_ptr):
        try:
            vin, vout = self._view(x_ptr), self._view(y_ptr)
            y = self.fn(vin)
            if not isinstance(y, torch.Tensor) or y.shape != vin.shape:
                raise ValueError("the operator must map a vector to a vector of the same shape")
            vout.copy_(y)
            self.calls += 1
            return 0
        except BaseException as e:  # never let an exception unwind through the C frames
            self.errors.append(e)
            return 1

    def close(self):
        if getattr(self, "_h", None):
            lib().hipk_csr_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def solve_matrix_free(kind: str, A_fn, b: torch.Tensor, x: torch.Tensor, *, tol: float, atol: float, maxiter: Optional[int],
                      restart: int = 20, solve_method: str = "batched", M=None, dinv: Optional[torch.Tensor] = None) -> SolveStats:
    """cg / bicgstab / gmres with a MATRIX-FREE operator (`_normalize_matvec` takes a callable for all three solvers,
    TSL:176-208) on the device-resident C loops: `A_fn` is called where the loops launch their SpMV (hipk_op_create), the fused
    vector kernels, the device stop word and the pacing are the matrix path's.  M: None, a callable (bicgstab / gmres: the
    *_solve_cb loops) or `dinv` for the Jacobi forms.  `x` holds x0 on entry and the solution on return."""
    assert b.is_cuda and b.is_contiguous() and x.is_contiguous() and x.dtype == b.dtype and x.shape == b.shape and b.ndim == 1
    h = OpHandle(A_fn, b.numel(), b.dtype, b.device)
    try:
        # what the callback's pointers may point into: b, x and the workspace (the solve functions register theirs)
        h.regions = [b, x]
        if dinv is not None:
            if kind == "gmres":
                st = solve_pgmres(h, dinv, b, x, tol=tol, atol=atol, maxiter=maxiter, restart=restart, solve_method=solve_method)
            else:
                st = solve_pcg(h, dinv, b, x, tol=tol, atol=atol, maxiter=maxiter, method=kind)
        elif M is not None:
            if kind == "cg":
                raise HipkError("solve_matrix_free: cg with a callable M runs on solve_cg_stepwise")
            st = _solve_with_callback(kind, h, M, b, x, tol=tol, atol=atol, maxiter=maxiter, restart=restart,
                                      solve_method=solve_method)
        else:
            st = solve(kind, h, b, x, tol=tol, atol=atol, maxiter=maxiter, restart=restart, solve_method=solve_method)
        if h.errors:
            torch.cuda.synchronize(b.device)
            raise h.errors[0]
        st.method = f"{kind}_matrix_free" + ("_jacobi" if dinv is not None else "_callable_M" if M is not None else "")
        return st
    except HipkError:
        if h.errors:
            torch.cuda.synchronize(b.device)
            raise h.errors[0]
        raise
    finally:
        torch.cuda.synchronize(b.device)   # nothing of the solve may still read the handle's scratch
        h.close()


def solve_bicgstab_callable(h: CsrHandle, M, b: torch.Tensor, x: torch.Tensor, *, tol: float, atol: float,
                            maxiter: Optional[int], check_every: int = 0) -> SolveStats:
    """hipk_pbicgstab_solve_cb: the device-resident BiCGStab loop with a CALLABLE preconditioner (TSL:859-964 with M)."""
    return _solve_with_callback("bicgstab", h, M, b, x, tol=tol, atol=atol, maxiter=maxiter, check_every=check_every)


def solve_gmres_callable(h: CsrHandle, M, b: torch.Tensor, x: torch.Tensor, *, tol: float, atol: float,
                         maxiter: Optional[int], restart: int = 20, solve_method: str = "batched") -> SolveStats:
    """hipk_pgmres_solve_cb: GMRES with a CALLABLE preconditioner applied after every A (TSL:641-803 with M)."""
    return _solve_with_callback("gmres", h, M, b, x, tol=tol, atol=atol, maxiter=maxiter, restart=restart,
                                solve_method=solve_method)


def _solve_with_callback(kind: str, h: CsrHandle, M, b: torch.Tensor, x: torch.Tensor, *, tol: float, atol: float,
                         maxiter: Optional[int], check_every: int = 0, restart: int = 20,
                         solve_method: str = "batched") -> SolveStats:
    """The C loop calls back here where the reference applies M -- BiCGStab: phat = M(p), shat = M(s), M(b - A x);
    GMRES: M(A v), M(b - A x), M b -- with pointers into the workspace; they are wrapped as views of the workspace
    tensor (no copy in), `M` runs on the current stream, its result is copied into the output slot.  No
    synchronisation inside the loop."""
    if h.shape[0] != h.shape[1]:
        raise ValueError(f"linear operator must be a square matrix, but has shape: {h.shape}")
    for t in (b, x):
        assert t.is_contiguous() and t.dtype == h.dtype and t.numel() == h.n and t.device == h.device
    L = lib()
    prm = Params()
    prm.tol, prm.atol = float(tol), float(atol)
    prm.maxiter = -1 if maxiter is None else int(maxiter)
    prm.check_every = int(check_every)
    prm.gpu_tolerances = 1
    if kind == "gmres":
        prm.restart = int(restart)
        prm.gmres_method = {"batched": GMRES_BATCHED, "incremental": GMRES_INCREMENTAL}[solve_method]
        wb = int(L.hipk_gmres_work_bytes(h.n, int(restart), _dtype_code(h.dtype)))
        fn = L.hipk_pgmres_solve_cb
    else:
        wb = int(L.hipk_pbicgstab_work_bytes(h.n, _dtype_code(h.dtype)))
        fn = L.hipk_pbicgstab_solve_cb
    work = torch.empty(wb, dtype=torch.uint8, device=h.device)
    if hasattr(h, "regions"):   # OpHandle: the operator callback resolves raw pointers against these tensors
        h.regions.append(work)
    base, nbytes = work.data_ptr(), h.n * work.new_empty(0, dtype=h.dtype).element_size()
    errors = []

    def view(ptr):
        off = int(ptr) - base
        if off < 0 or off + nbytes > wb:
            raise HipkError("preconditioner callback: pointer outside the workspace")
        return work[off:off + nbytes].view(h.dtype)

    def callback(_user, in_ptr, out_ptr):
        try:
            vin, vout = view(in_ptr), view(out_ptr)
            z = M(vin)
            if not isinstance(z, torch.Tensor) or z.shape != vin.shape:
                raise ValueError("the preconditioner must map a vector to a vector of the same shape")
            vout.copy_(z)
            return 0
        except BaseException as e:  # never let an exception unwind through the C frames
            errors.append(e)
            return 1
    cb = PRECOND_FN(callback)
    st = Stats()
    with h._lock, torch.cuda.device(h.device):
        rc = fn(h.ptr, cb, None, b.data_ptr(), x.data_ptr(), work.data_ptr(), wb, ctypes.byref(prm), ctypes.byref(st),
                _stream(h.device))
    if errors:
        torch.cuda.synchronize(h.device)
        raise errors[0]
    _check(rc, f"hipk_p{kind}_solve_cb")
    return SolveStats(method=f"{kind}_callable_M", iterations=st.iterations, matvecs=st.matvecs, info=st.info,
                      breakdown=st.breakdown, b_norm=st.b_norm, residual_norm=st.residual_norm, x_norm=st.x_norm,
                      threshold=st.threshold, recurrence_rs=st.recurrence_rs, solve_ms=st.solve_ms,
                      spmv_ms_avg=st.spmv_ms_avg, spmv_profiled=st.spmv_profiled,
                      dispatch_span_ms_avg=st.dispatch_span_ms_avg)
