"""the lid-driven-cavity caller (examples/ldc_projection.py) on the CSR path.  The pressure right-hand
sides of its first three time steps are compared with the ones the REFERENCE's stepper produced (recorded in
tests/golden/ldc_nx*_step*.npz by oracle/gen_golden.py: Re = 100, gmres(tol=1e-10, maxiter=1000, restart=30))."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-sparse-linalg-torch-amgx.cg.bicg.gmres_amd", "examples"))
GOLD = os.path.join(ROOT, "tests", "golden")


def _run(nx, device, steps=3, **kw):
    from ldc_projection import LidDrivenCavity
    sim = LidDrivenCavity(nx=nx, Re=100.0, method="gmres", device=device, **kw)
    reps = [sim.step() for _ in range(steps)]
    return sim, reps


@pytest.mark.parametrize("nx", [8, 16, 32])
def test_pressure_right_hand_sides_match_the_reference_stepper_cpu(nx):
    sim, reps = _run(nx, "cpu")
    for k in range(3):
        d = np.load(os.path.join(GOLD, f"ldc_nx{nx}_step{k}.npz"))
        b = sim.rhs_log[k].numpy()
        assert np.abs(b - d["b"]).max() <= 1e-7 * np.abs(d["b"]).max(), (nx, k)
        assert reps[k].info == 0
    A = sim.A
    assert np.array_equal(A.crow_indices().numpy(), d["crow"]) and np.array_equal(A.col_indices().numpy(), d["col"])
    assert np.array_equal(A.values().numpy(), d["val"])


@pytest.mark.gpu
@pytest.mark.parametrize("nx", [8, 16, 32])
def test_pressure_right_hand_sides_match_the_reference_stepper_gpu(hipk, nx):
    sim, reps = _run(nx, "cuda:0")
    for k in range(3):
        d = np.load(os.path.join(GOLD, f"ldc_nx{nx}_step{k}.npz"))
        b = sim.rhs_log[k].cpu().numpy()
        assert np.abs(b - d["b"]).max() <= 1e-7 * np.abs(d["b"]).max(), (nx, k)
        assert reps[k].info == 0


@pytest.mark.gpu
def test_one_handle_serves_every_time_step_and_warm_start_saves_work(hipk):
    from pytorch_sparse_solver import _hipk
    sim, reps = _run(100, "cuda:0", steps=12)
    h = _hipk.handle_for(sim.A)
    assert h.path() == "coded"                                   # the constant-coefficient pressure matrix
    assert _hipk.handle_for(sim.A) is h                          # cached by storage identity: built once
    assert all(r.info == 0 for r in reps) and reps[-1].mass_residual < 1e-6
    cold = sum(r.matvecs for r in reps[2:])
    sim2, reps2 = _run(100, "cuda:0", steps=12, warm_start=True)
    warm = sum(r.matvecs for r in reps2[2:])
    assert all(r.info == 0 for r in reps2) and warm < cold
