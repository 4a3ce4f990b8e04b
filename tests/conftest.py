import json
import os
import sys
import warnings

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)
warnings.filterwarnings("ignore", message=".*Sparse CSR tensor support is in beta.*")

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_index():
    with open(os.path.join(GOLDEN, "index.json")) as f:
        return json.load(f)


def golden_runs(solver=None):
    runs = golden_index()["runs"]
    return [r for r in runs if solver is None or r["solver"] == solver]


def ldc100_runs():
    """BASELINE config 4 at the reference's default size (nx = 100, Re = 400), oracle/gen_golden_r2.py."""
    with open(os.path.join(GOLDEN, "ldc100_index.json")) as f:
        return json.load(f)["runs"]


def gmres_tol_runs():
    with open(os.path.join(GOLDEN, "gmres_tol.json")) as f:
        return json.load(f)["runs"]


def load_case(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def run_id(r):
    return f"{r['case']}-{r['tag']}"


# BiCGStab is trajectory-chaotic: a different (equally valid)
# summation order moves the iteration count by a few percent, and on the ill-conditioned
# tridiagonal case at tol=1e-10 the recurrence residual and the true residual part ways.
BICGSTAB_MATVEC_BAND = 0.15


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def hipk():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from pytorch_sparse_solver import _hipk
    _hipk.lib()
    return _hipk
