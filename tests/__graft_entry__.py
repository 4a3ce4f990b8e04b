"""Driver entry points: build() compiles everything (no GPU needed), smoke() runs one small
solve of the hot path on cuda:0 and checks it against the oracle."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def build() -> None:
    """hipcc --offload-arch=gfx950 for libhipk.so (in-tree), gcc for the oracle; import the package."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(PKG, "csrc"), "-j4"])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    import torch  # noqa: F401  (must precede loading libhipk.so)
    import pytorch_sparse_solver  # noqa: F401
    from pytorch_sparse_solver import _hipk
    L = _hipk.lib()
    missing = [s for s in _hipk.SYMBOLS if not hasattr(L, s)]
    if missing:
        raise RuntimeError(f"libhipk.so lacks symbols: {missing}")
    print(f"built {_hipk.LIB_PATH} (hipk_version={L.hipk_version()}) and the oracle")


def smoke() -> None:
    """One small CG solve (64x64 Poisson) on cuda:0 through the public API; must be bit-identical
    to the CPU oracle and converge like the reference fixture says (103 operator applications)."""
    import numpy as np
    import torch
    from oracle import oracle as O
    from pytorch_sparse_solver import _hipk
    from pytorch_sparse_solver.module_a import cg, get_last_stats
    from pytorch_sparse_solver.utils.matrix_utils import create_poisson_2d_csr

    if not torch.cuda.is_available():
        raise RuntimeError("smoke() needs a GPU")
    if _hipk.lib().hipk_device_count() < 1:
        raise RuntimeError("no gfx950 device visible to libhipk.so")
    nx = 64
    A = create_poisson_2d_csr(nx, nx, device="cuda:0")
    b = torch.ones(nx * nx, dtype=torch.float64, device="cuda:0")
    x, info = cg(A, b, tol=1e-6)
    st = get_last_stats()
    Ac = A.cpu()
    ref = O.cg(Ac.crow_indices().numpy(), Ac.col_indices().numpy(), Ac.values().numpy(), np.ones(nx * nx), tol=1e-6)
    assert info == 0 and ref.info == 0, (info, ref.info)
    assert st.matvecs == ref.matvecs == 103, (st.matvecs, ref.matvecs)
    assert np.array_equal(x.cpu().numpy(), ref.x), "HIP CG differs from the oracle"
    print(f"smoke ok: cg 64x64 Poisson, {st.iterations} iterations, relres "
          f"{st.residual_norm / st.b_norm:.3e}, bit-identical to the oracle, {st.solve_ms:.3f} ms")


if __name__ == "__main__":
    build()
    if len(sys.argv) > 1 and sys.argv[1] == "smoke":
        smoke()
