"""Backend availability (reference: utils/availability.py:13-119).

Only Module A exists in this MI355X build.  Module B (pyamgx/AMGX) and Module C
(torch.sparse.spsolve/cuDSS) are NVIDIA-only backends outside the hot path, so they
report False and the dispatcher rejects them exactly as the reference does on a
machine without them (solver.py:219-225).
"""
from functools import lru_cache
from typing import Dict, List


@lru_cache(maxsize=1)
def check_module_a_available() -> bool:
    try:
        import torch  # noqa: F401
        return True
    except ImportError:
        return False


@lru_cache(maxsize=1)
def check_module_b_available() -> bool:
    return False


@lru_cache(maxsize=1)
def check_module_c_available() -> bool:
    return False


def get_available_backends() -> Dict[str, bool]:
    return {
        'module_a': check_module_a_available(),
        'module_b': check_module_b_available(),
        'module_c': check_module_c_available(),
    }


def get_available_backend_list() -> List[str]:
    return [k for k, v in get_available_backends().items() if v]


def hip_extension_status() -> Dict[str, object]:
    """New: whether the gfx950 extension is built and a device is visible."""
    import os
    from .. import _hipk
    st = {'library': _hipk.LIB_PATH, 'built': os.path.exists(_hipk.LIB_PATH), 'gfx950_devices': 0}
    if st['built']:
        try:
            st['gfx950_devices'] = int(_hipk.lib().hipk_device_count())
        except Exception as e:  # pragma: no cover
            st['error'] = str(e)
    return st


def print_availability_report() -> None:
    b = get_available_backends()
    print("=" * 60)
    print("PyTorch Sparse Solver (MI355X build) - Module Availability Report")
    print("=" * 60)
    print(f"Module A (CG / BiCGStab / GMRES, gfx950 kernels): {'available' if b['module_a'] else 'NOT available'}")
    print("Module B (pyamgx): not part of this build")
    print("Module C (cuDSS): not part of this build")
    print(f"HIP extension: {hip_extension_status()}")
    print("=" * 60)
