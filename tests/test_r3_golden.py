"""Round-3 fixtures (oracle/gen_golden_r3.py, the reference run in the build container):
 * BASELINE config 1 at its stated size -- the reference's own recipe (test_module_a.py:39-42) at n = 1000 as a WHOLE cg() solve:
   CPU generic path (dense and CSR), the oracle, and on the GPU the HIP path (dense-as-CSR = the row-per-wavefront SpMV);
 * GMRES with restart 40 and 64 (the reference accepts any restart, TSL:641-644): the oracle, the generic path, the HIP path."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_case
from pytorch_sparse_solver.module_a import cg, get_last_stats, gmres


def _index():
    with open(os.path.join(GOLDEN, "r3_index.json")) as f:
        return json.load(f)


def _config1():
    d = load_case("config1_spd_n1000")
    n = d["Gq"].shape[0]
    Gq = torch.from_numpy(d["Gq"].astype(np.int64))
    # exact in fp64 whatever the summation order (multiples of 1/256): the matrix of the fixture, bit for bit
    A = (Gq @ Gq.T).to(torch.float64) / 256 + n * torch.eye(n, dtype=torch.float64)
    return d, A, torch.from_numpy(d["b"])


@pytest.mark.parametrize("form", ["dense", "csr"])
def test_config1_whole_solve_generic_cpu_path(form):
    d, A, b = _config1()
    r = _index()["config1"]["runs"][form]
    x, info = cg(A if form == "dense" else A.to_sparse_csr(), b, tol=1e-6)
    st = get_last_stats()
    x_ref = d["x_" + form]
    assert info == r["info"] == 0 and st.matvecs == r["matvecs"]
    assert np.linalg.norm(x.numpy() - x_ref) <= 1e-8 * np.linalg.norm(x_ref)
    assert torch.norm(b - A @ x) <= 1e-6 * torch.norm(b)


def test_config1_whole_solve_oracle(oracle):
    d, A, b = _config1()
    r = _index()["config1"]["runs"]["csr"]
    Ac = A.to_sparse_csr()
    res = oracle.cg(Ac.crow_indices().numpy().astype(np.int32), Ac.col_indices().numpy().astype(np.int32),
                    Ac.values().numpy(), b.numpy(), tol=1e-6)
    assert res.info == r["info"] and res.matvecs == r["matvecs"]
    assert np.linalg.norm(res.x - d["x_csr"]) <= 1e-8 * np.linalg.norm(d["x_csr"])


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["dense", "csr"])
def test_config1_whole_solve_hip_path(oracle, hipk, form):
    """n = 1000, 1000 entries per row: the row-per-wavefront SpMV inside the device-resident CG loop; the reference's count and
    solution, and the oracle's bits."""
    d, A, b = _config1()
    r = _index()["config1"]["runs"][form]
    Ad = A.cuda() if form == "dense" else A.to_sparse_csr().cuda()
    x, info = cg(Ad, b.cuda(), tol=1e-6)
    st = get_last_stats()
    assert info == r["info"] == 0 and st.matvecs == r["matvecs"]
    x_ref = d["x_" + form]
    assert np.linalg.norm(x.cpu().numpy() - x_ref) <= 1e-8 * np.linalg.norm(x_ref)
    Ac = A.to_sparse_csr()
    res = oracle.cg(Ac.crow_indices().numpy().astype(np.int32), Ac.col_indices().numpy().astype(np.int32),
                    Ac.values().numpy(), b.numpy(), tol=1e-6)
    assert np.array_equal(x.cpu().numpy(), res.x)


def _csr(d):
    n = int(d["n"])
    return torch.sparse_csr_tensor(torch.from_numpy(d["crow"]).long(), torch.from_numpy(d["col"]).long(),
                                   torch.from_numpy(d["val"]), size=(n, n))


BIG = _index()["big_restart"]


@pytest.mark.parametrize("r", BIG, ids=lambda r: f"{r['case']}-{r['tag']}")
def test_big_restart_oracle_reproduces_reference(oracle, r):
    d, xs = load_case(r["case"]), load_case(f"r3_{r['case']}_bigrestart")
    res = oracle.gmres(d["crow"], d["col"], d["val"], d["b"], **r["kwargs"])
    x_ref = xs[r["tag"] + "_x"]
    assert res.info == r["info"] and res.matvecs == r["matvecs"]
    assert np.linalg.norm(res.x - x_ref) <= 1e-8 * np.linalg.norm(x_ref)


@pytest.mark.parametrize("r", BIG, ids=lambda r: f"{r['case']}-{r['tag']}")
def test_big_restart_generic_path_reproduces_reference(r):
    d, xs = load_case(r["case"]), load_case(f"r3_{r['case']}_bigrestart")
    x, info = gmres(_csr(d), torch.from_numpy(d["b"]), **r["kwargs"])
    x_ref = xs[r["tag"] + "_x"]
    assert info == r["info"] and get_last_stats().matvecs == r["matvecs"]
    assert np.linalg.norm(x.numpy() - x_ref) <= 1e-8 * np.linalg.norm(x_ref)


@pytest.mark.gpu
@pytest.mark.parametrize("r", BIG, ids=lambda r: f"{r['case']}-{r['tag']}")
def test_big_restart_hip_path_bitwise_vs_oracle(oracle, hipk, r):
    """restart 40 / 64 stay on the HIP kernels (no generic-path warning) and reproduce the oracle bit for bit."""
    import warnings
    d, xs = load_case(r["case"]), load_case(f"r3_{r['case']}_bigrestart")
    A, b = _csr(d).cuda(), torch.from_numpy(d["b"]).cuda()
    with warnings.catch_warnings():
        warnings.simplefilter("error", RuntimeWarning)     # the "leaves the HIP path" warning would be one
        x, info = gmres(A, b, **r["kwargs"])
    st = get_last_stats()
    res = oracle.gmres(d["crow"], d["col"], d["val"], d["b"], gpu_tolerances=True, **r["kwargs"])
    assert info == res.info and st.matvecs == res.matvecs
    assert np.array_equal(x.cpu().numpy(), res.x)
    # the fixture was produced on the reference's cpu tolerance branch: looser or equal on the device branch
    assert info == r["info"] and st.matvecs <= r["matvecs"]
    x_ref = xs[r["tag"] + "_x"]
    xg = x.cpu().numpy()
    if r["case"].startswith("ldc"):
        xg, x_ref = xg - xg.mean(), x_ref - x_ref.mean()
    assert np.linalg.norm(xg - x_ref) <= 1e-5 * np.linalg.norm(x_ref)
