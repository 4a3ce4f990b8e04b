"""GPU parity of the CODED SpMV path (csrc/hipk_coded.h: one byte per entry for matrices with at most 256 distinct
(col - row, value) pairs) -- against the oracle and against the plain CSR kernels on the SAME handle, bit for bit --
and of the structure analysis that selects it."""
import os

import numpy as np
import pytest
import torch

from conftest import load_case

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def banded(n, offsets, values_of, seed=0):
    """CSR of a banded matrix: row i has an entry at column i + o for every offset o inside the matrix; the
    value is values_of(i, k) for the k-th offset."""
    offsets = np.sort(np.asarray(offsets))
    rows = np.repeat(np.arange(n), len(offsets))
    cols = rows + np.tile(offsets, n)
    kk = np.tile(np.arange(len(offsets)), n)
    keep = (cols >= 0) & (cols < n)
    rows, cols, kk = rows[keep], cols[keep], kk[keep]
    crow = np.zeros(n + 1, dtype=np.int64)
    np.add.at(crow, rows + 1, 1)
    crow = np.cumsum(crow)
    val = values_of(rows, kk).astype(np.float64)
    return crow, cols.astype(np.int64), val


def make_handle(hipk, crow, col, val, n, dtype=torch.float64):
    return hipk.CsrHandle(torch.from_numpy(crow).to(DEV), torch.from_numpy(col).to(DEV),
                          torch.from_numpy(val).to(DEV).to(dtype), (n, n))


def both_paths(hipk, h, x, expect="coded"):
    assert h.path() == expect
    y_coded = hipk.spmv(h, x).cpu().numpy()
    h.set_path(plain_only=True)
    assert h.path() in ("tile_fast", "tile")
    y_plain = hipk.spmv(h, x).cpu().numpy()
    h.set_path(plain_only=False)
    return y_coded, y_plain


@pytest.mark.parametrize("case", ["poisson_nx64", "convdiff_nx64", "ldc_nx32_step0", "poisson_nx8"])
def test_fixture_stencils_take_the_coded_path_and_match(hipk, oracle, case):
    d = load_case(case)
    n = int(d["n"])
    h = make_handle(hipk, d["crow"], d["col"], d["val"], n)
    x = np.random.default_rng(1).standard_normal(n)
    y_coded, y_plain = both_paths(hipk, h, torch.from_numpy(x).to(DEV))
    ref = oracle.spmv(d["crow"], d["col"], d["val"], x)
    assert np.array_equal(y_coded, ref) and np.array_equal(y_plain, ref)
    assert h.format_bytes() < h.spmv_bytes()


@pytest.mark.parametrize("layout", ["sell", "sell-x2", "csr"])
@pytest.mark.parametrize("offsets", [[-300, -1, 0, 1, 300], [-1, 0, 1], [-40, -7, -1, 0, 1, 7, 40], [0, 5, 9, 11, 50, 51]])
@pytest.mark.parametrize("n", [1, 255, 256, 257, 1023, 70_001, 1_200_011])
def test_coded_tile_boundaries_layouts_and_widths(hipk, oracle, n, offsets, layout, monkeypatch):
    monkeypatch.setenv("HIPK_SPMV_CODED_LAYOUT", layout.split("-")[0])
    monkeypatch.setenv("HIPK_SPMV_SELL_LOOP", "2" if layout == "sell-x2" else "1")   # grid = 2 x resident workgroups
    vals = np.array([-1.0, -1.5, 4.0, -1.25, -1.75, 0.5, 3.0])
    crow, col, val = banded(n, offsets, lambda r, k: vals[k])
    h = make_handle(hipk, crow, col, val, n)
    x = np.random.default_rng(n).standard_normal(n)
    y_coded, y_plain = both_paths(hipk, h, torch.from_numpy(x).to(DEV))
    ref = oracle.spmv(crow, col, val, x)
    assert np.array_equal(y_coded, ref) and np.array_equal(y_plain, ref)


@pytest.mark.parametrize("layout", ["sell", "csr"])
def test_coded_wide_stencil_more_than_4096_codes_per_tile(hipk, oracle, layout, monkeypatch):
    monkeypatch.setenv("HIPK_SPMV_CODED_LAYOUT", layout)
    n = 5000
    offs = list(range(-13, 14))                                   # 27 entries per row: 6912 codes per tile
    crow, col, val = banded(n, offs, lambda r, k: (k - 13.0) / 7.0 + 3.0 * (k == 13))
    h = make_handle(hipk, crow, col, val, n)
    x = np.random.default_rng(3).standard_normal(n)
    y_coded, y_plain = both_paths(hipk, h, torch.from_numpy(x).to(DEV))
    ref = oracle.spmv(crow, col, val, x)
    assert np.array_equal(y_coded, ref) and np.array_equal(y_plain, ref)


def test_dictionary_capacity_256_pairs_yes_257_no(hipk, oracle):
    n = 4096
    # 4 offsets x 64 values per offset = 256 distinct pairs (interior rows carry all of them)
    crow, col, val = banded(n, [-2, -1, 1, 2], lambda r, k: 1.0 + (r % 64) + 100.0 * k)
    h = make_handle(hipk, crow, col, val, n)
    assert h.path() == "coded"
    x = np.random.default_rng(4).standard_normal(n)
    assert np.array_equal(hipk.spmv(h, torch.from_numpy(x).to(DEV)).cpu().numpy(), oracle.spmv(crow, col, val, x))
    val2 = val.copy()
    val2[len(val2) // 2] = 12345.678                               # one more pair
    h2 = make_handle(hipk, crow, col, val2, n)
    assert h2.path() != "coded"
    assert np.array_equal(hipk.spmv(h2, torch.from_numpy(x).to(DEV)).cpu().numpy(), oracle.spmv(crow, col, val2, x))


def test_ragged_rows_with_few_pairs_fall_back_to_the_csr_layout(hipk, oracle):
    """Row lengths 0..32 with one value per offset: the byte planes would be mostly padding (> 2 x nnz), so the
    codes stay in CSR order; results unchanged."""
    n = 6000
    rng = np.random.default_rng(12)
    lens = np.where(rng.random(n) < 0.9, rng.integers(0, 3, n), 32)
    crow = np.zeros(n + 1, dtype=np.int64)
    crow[1:] = np.cumsum(lens)
    col = np.concatenate([(np.arange(l) * 7 + r) % n for r, l in enumerate(lens)])
    order = np.concatenate([np.argsort(col[crow[r]:crow[r + 1]], kind="stable") + crow[r] for r in range(n)])
    col = col[order]
    off = col - np.repeat(np.arange(n), lens)
    val = 1.0 + (off % 5)
    h = make_handle(hipk, crow, col.astype(np.int64), val.astype(np.float64), n)
    assert h.path() == "coded"
    x = rng.standard_normal(n)
    y_coded, y_plain = both_paths(hipk, h, torch.from_numpy(x).to(DEV))
    ref = oracle.spmv(crow, col, val, x)
    assert np.array_equal(y_coded, ref) and np.array_equal(y_plain, ref)


def test_random_values_and_long_rows_are_not_coded(hipk, monkeypatch):
    n = 3000
    crow, col, val = banded(n, [-1, 0, 1], lambda r, k: np.random.default_rng(0).standard_normal(len(r)))
    assert make_handle(hipk, crow, col, val, n).path() == "offset_coded"      # few offsets, arbitrary values
    monkeypatch.setenv("HIPK_SPMV_OFFSET_CODED", "0")
    assert make_handle(hipk, crow, col, val, n).path() == "tile_fast"
    monkeypatch.delenv("HIPK_SPMV_OFFSET_CODED")
    rng = np.random.default_rng(1)                                            # random columns: > 255 distinct offsets
    lens = np.full(n, 4)
    crow = np.concatenate([[0], np.cumsum(lens)])
    col = np.concatenate([np.sort(rng.choice(n, 4, replace=False)) for _ in range(n)])
    assert make_handle(hipk, crow, col, rng.standard_normal(4 * n), n).path() == "tile_fast"
    crow, col, val = banded(n, list(range(-20, 21)), lambda r, k: 1.0 + k)       # 41 entries per row > 32
    assert make_handle(hipk, crow, col, val, n).path() == "tile"


def test_signed_zero_and_nan_payload_are_distinct_pairs(hipk, oracle):
    n = 2000
    crow, col, val = banded(n, [-1, 0, 1], lambda r, k: np.array([0.0, 2.0, -0.0])[k])
    assert np.signbit(val).any()
    h = make_handle(hipk, crow, col, val, n)
    assert h.path() == "coded"
    x = np.random.default_rng(6).standard_normal(n)
    x[::7] = -x[::7]
    y = hipk.spmv(h, torch.from_numpy(x).to(DEV)).cpu().numpy()
    ref = oracle.spmv(crow, col, val, x)
    assert np.array_equal(np.signbit(y), np.signbit(ref)) and np.array_equal(y, ref)


def test_environment_switch_disables_the_coded_form(hipk, monkeypatch):
    d = load_case("poisson_nx64")
    monkeypatch.setenv("HIPK_SPMV_CODED", "0")
    h = make_handle(hipk, d["crow"], d["col"], d["val"], int(d["n"]))
    assert h.path() == "tile_fast" and h.format_bytes() == h.spmv_bytes()


@pytest.mark.parametrize("layout", ["sell", "csr"])
def test_coded_fp32_storage(hipk, oracle, layout, monkeypatch):
    monkeypatch.setenv("HIPK_SPMV_CODED_LAYOUT", layout)
    d = load_case("convdiff_nx64")
    n = int(d["n"])
    v32 = d["val"].astype(np.float32)
    h = make_handle(hipk, d["crow"], d["col"], d["val"], n, dtype=torch.float32)
    x = np.random.default_rng(8).standard_normal(n).astype(np.float32)
    y_coded, y_plain = both_paths(hipk, h, torch.from_numpy(x).to(DEV))
    ref = oracle.spmv32(d["crow"], d["col"], v32, x)
    assert np.array_equal(y_coded, ref) and np.array_equal(y_plain, ref)


def test_coded_fused_dots_and_residual_form(hipk, oracle):
    """spmv_ex modes on the coded path: <w, out>, <out, out>, out = b - A x; chunk partials equal the plain path's."""
    from pytorch_sparse_solver.utils.matrix_utils import create_poisson_2d_csr
    A = create_poisson_2d_csr(300, 300, device=DEV)
    h = hipk.handle_for(A)
    n = 90_000
    g = torch.Generator(device=DEV).manual_seed(0)
    x = torch.randn(n, dtype=torch.float64, device=DEV, generator=g)
    w = torch.randn(n, dtype=torch.float64, device=DEV, generator=g)
    b = torch.randn(n, dtype=torch.float64, device=DEV, generator=g)
    L = hipk.lib()
    G = int(L.hipk_chunk_count(n))
    s = torch.cuda.current_stream().cuda_stream
    outs = {}
    for plain in (False, True):
        h.set_path(plain_only=plain)
        assert h.path() == ("tile_fast" if plain else "coded")
        y = torch.empty_like(x)
        p0 = torch.zeros(G, dtype=torch.float64, device=DEV)
        p1 = torch.zeros(G, dtype=torch.float64, device=DEV)
        hipk._check(L.hipk_spmv_ex(h._h, x.data_ptr(), y.data_ptr(), 7, w.data_ptr(), b.data_ptr(), p0.data_ptr(),
                                   p1.data_ptr(), None, 0, s), "hipk_spmv_ex")
        outs[plain] = (y.cpu().numpy(), p0.cpu().numpy(), p1.cpu().numpy())
    h.set_path(plain_only=False)
    for a, c in zip(outs[False], outs[True]):
        assert np.array_equal(a, c)
    crow, col, val = (t.cpu().numpy() for t in (A.crow_indices(), A.col_indices(), A.values()))
    y_ref = b.cpu().numpy() - oracle.spmv(crow, col, val, x.cpu().numpy())
    assert np.array_equal(outs[False][0], y_ref)


@pytest.mark.parametrize("solver", ["cg", "bicgstab", "gmres"])
def test_whole_solves_identical_on_both_paths(hipk, solver):
    from pytorch_sparse_solver.utils.matrix_utils import create_convdiff_2d_csr, create_poisson_2d_csr
    A = create_poisson_2d_csr(150, 150, device=DEV) if solver == "cg" else create_convdiff_2d_csr(150, 150, device=DEV)
    h = hipk.handle_for(A)
    n = A.shape[0]
    b = torch.randn(n, dtype=torch.float64, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1))
    res = {}
    for plain in (False, True):
        h.set_path(plain_only=plain)
        x = torch.zeros_like(b)
        st = hipk.solve(solver, h, b, x, tol=1e-8, atol=0.0, maxiter=None if solver != "gmres" else 200, restart=20)
        res[plain] = (x.cpu().numpy(), st.iterations, st.info, st.residual_norm)
    h.set_path(plain_only=False)
    assert np.array_equal(res[False][0], res[True][0]) and res[False][1:] == res[True][1:]
    assert res[False][2] == 0


def _spmv_ex_all_modes(hipk, h, x, w, b):
    L = hipk.lib()
    G = int(L.hipk_chunk_count(x.numel()))
    y = torch.empty_like(x)
    p0 = torch.zeros(G, dtype=torch.float64, device=DEV)
    p1 = torch.zeros(G, dtype=torch.float64, device=DEV)
    hipk._check(L.hipk_spmv_ex(h._h, x.data_ptr(), y.data_ptr(), 7, w.data_ptr(), b.data_ptr(), p0.data_ptr(),
                               p1.data_ptr(), None, 0, torch.cuda.current_stream().cuda_stream), "hipk_spmv_ex")
    return y.cpu().numpy(), p0.cpu().numpy(), p1.cpu().numpy()


@pytest.mark.parametrize("chunked", ["1", "0"])
def test_full_size_poisson_coded_equals_plain(hipk, chunked, monkeypatch):
    """BASELINE config 2 matrix (N = 4M): y and the chunk partials of both fused dots are the same bits on the plain
    path, on the persistent coded kernel with a combine launch (chunked=0) and on its chunk-per-workgroup form
    that folds the partials itself (chunked=1); the coded form streams < 30 % of the bytes."""
    from pytorch_sparse_solver.utils.matrix_utils import create_poisson_2d_csr
    monkeypatch.setenv("HIPK_SPMV_SELL_CHUNKED", chunked)
    nx = 2000
    A = create_poisson_2d_csr(nx, nx, device=DEV)
    h = hipk.CsrHandle(A.crow_indices(), A.col_indices(), A.values(), A.shape)
    g = torch.Generator(device=DEV).manual_seed(2)
    x, w, b = (torch.randn(nx * nx, dtype=torch.float64, device=DEV, generator=g) for _ in range(3))
    assert h.path() == "coded" and h.format_bytes() < 0.3 * h.spmv_bytes()
    coded = _spmv_ex_all_modes(hipk, h, x, w, b)
    h.set_path(plain_only=True)
    plain = _spmv_ex_all_modes(hipk, h, x, w, b)
    for a, c in zip(coded, plain):
        assert np.array_equal(a, c)


def test_chunked_form_with_a_ragged_last_chunk_and_larger_chunks(hipk, oracle):
    """n = 5M + 77 rows: chunk size 4096 (16 tiles per chunk), last chunk and last tile partial."""
    n = 5_000_077
    crow, col, val = banded(n, [-1500, -1, 0, 1, 1500], lambda r, k: np.array([-1.0, -1.0, 4.0, -1.0, -1.0])[k])
    h = make_handle(hipk, crow, col, val, n)
    g = torch.Generator(device=DEV).manual_seed(3)
    x, w, b = (torch.randn(n, dtype=torch.float64, device=DEV, generator=g) for _ in range(3))
    assert h.path() == "coded"
    coded = _spmv_ex_all_modes(hipk, h, x, w, b)
    h.set_path(plain_only=True)
    plain = _spmv_ex_all_modes(hipk, h, x, w, b)
    for a, c in zip(coded, plain):
        assert np.array_equal(a, c)
    assert np.array_equal(coded[0], b.cpu().numpy() - oracle.spmv(crow, col, val, x.cpu().numpy()))


def _spmv_ex_mode(hipk, h, mode, x, w, b):
    L = hipk.lib()
    G = int(L.hipk_chunk_count(x.numel()))
    y = torch.empty_like(x)
    p0 = torch.zeros(G, dtype=torch.float64, device=DEV)
    p1 = torch.zeros(G, dtype=torch.float64, device=DEV)
    hipk._check(L.hipk_spmv_ex(h._h, x.data_ptr(), y.data_ptr(), mode, w.data_ptr(), b.data_ptr(), p0.data_ptr(),
                               p1.data_ptr(), None, 0, torch.cuda.current_stream().cuda_stream), "hipk_spmv_ex")
    return y.cpu().numpy(), p0.cpu().numpy(), p1.cpu().numpy()


@pytest.mark.parametrize("offsets", [[-1, 0, 1], [-1500, -1, 0, 1], [-1500, -1, 0, 1, 1500], [-1500, -1, 1, 1500],
                                     [-9000, -1500, -1, 0, 1, 1500, 9000], [-9000, -1500, -2, -1, 0, 1, 1500, 9000]])
@pytest.mark.parametrize("strided", ["0", "1"])
@pytest.mark.parametrize("masked", ["0", "1"])
def test_uniform_tiles_two_rows_per_lane(hipk, oracle, offsets, strided, masked, monkeypatch):
    """strided = 1: the kernel's grouped walk (one workgroup per 4 consecutive tiles on an ordinary grid, tile sums through
    hipk_tile_combine_kernel) -- what row blocks of few large chunks and systems of N > 16 M take -- forced here at a size the
    CPU oracle checks in seconds.
    hipk_spmv_sell_wide_kernel (chunk-per-workgroup sizes, fp64, most tiles uniform): uniform tiles from 16-byte accesses, two
    rows per lane, the fused dots' 64-row sums on that layout; tile widths 3-4, 5 and 7-8, a partial last tile, uniform tiles
    with padding (the first and last rows of the band lack entries), an offset list without a diagonal.  Every mode the solvers
    use -- compiled-in <w, y> with w == x (the CG loop: w taken from the diagonal entry's load) and w != x, ||y||^2, the
    residual form with both dots -- against the plain CSR kernels bit for bit, y against the oracle."""
    n = 2_200_077
    monkeypatch.setenv("HIPK_SPMV_SELL_STRIDED", strided)
    # masked = 1 (opt-in, round 3): tiles whose rows have SUBSETS of one pattern (the band's first / last rows, a partial pattern)
    # go through the two-rows-per-lane path as well, absent entries skipped by a per-row presence mask
    monkeypatch.setenv("HIPK_SPMV_MASKED", masked)
    crow, col, val = banded(n, offsets, lambda r, k: 1.5 + 0.25 * k)
    h = make_handle(hipk, crow, col, val, n)
    g = torch.Generator(device=DEV).manual_seed(5)
    x, w, b = (torch.randn(n, dtype=torch.float64, device=DEV, generator=g) for _ in range(3))
    assert h.path() == "coded"
    runs = [(0, x, w), (1, x, x), (1, x, w), (2, x, w), (7, x, w), (6, x, w), (3, x, w)]
    coded = [_spmv_ex_mode(hipk, h, m, xx, ww, b) for m, xx, ww in runs]
    assert hipk.CsrHandle.last_spmv_kernel().startswith("hipk_spmv_sell_wide_kernel"), hipk.CsrHandle.last_spmv_kernel()
    assert hipk.CsrHandle.last_spmv_kernel().endswith("," + strided + ">"), hipk.CsrHandle.last_spmv_kernel()
    h.set_path(plain_only=True)
    plain = [_spmv_ex_mode(hipk, h, m, xx, ww, b) for m, xx, ww in runs]
    assert hipk.CsrHandle.last_spmv_kernel().startswith("hipk_spmv_kernel")
    h.set_path(plain_only=False)
    for (m, _, _), cr, pr in zip(runs, coded, plain):
        assert np.array_equal(cr[0], pr[0]), m
        if m & 1:
            assert np.array_equal(cr[1], pr[1]), m
        if m & 2:
            assert np.array_equal(cr[2], pr[2]), m
    assert np.array_equal(coded[0][0], oracle.spmv(crow, col, val, x.cpu().numpy()))


@pytest.mark.parametrize("nx", [4000, 8000])
def test_many_grid_lines_per_chunk_coded_equals_plain(hipk, nx, monkeypatch):
    """BASELINE config 5's matrix on ONE device (nx = 8000: N = 64 M, reduction chunks of 128 tiles = four grid lines) and
    N = 16 M (chunks of 32 tiles, grouped walk forced): the two-rows-per-lane kernel's grouped walk; 25 CG iterations (fused
    <p, Ap>, the residual form at the end) bitwise equal to the general CSR kernels, which the small-size tests tie to the
    oracle; and the CG property r_k = b - A x_k to rounding, checked with the plain kernels' SpMV."""
    from pytorch_sparse_solver.utils.matrix_utils import create_poisson_2d_csr
    if nx < 8000:
        monkeypatch.setenv("HIPK_SPMV_SELL_STRIDED", "1")                 # nx = 8000 takes it by default
    A = create_poisson_2d_csr(nx, nx, device=DEV)
    n = nx * nx
    h = hipk.CsrHandle(A.crow_indices(), A.col_indices(), A.values(), A.shape)
    assert h.path() == "coded"
    b = torch.ones(n, dtype=torch.float64, device=DEV)
    res = {}
    for plain in (False, True):
        h.set_path(plain_only=plain)
        x = torch.zeros_like(b)
        st = hipk.solve("cg", h, b, x, tol=1e-12, atol=0.0, maxiter=25)
        res[plain] = (x, st.iterations, st.info, st.residual_norm)
        if not plain:
            k = hipk.CsrHandle.last_spmv_kernel()
            assert k.startswith("hipk_spmv_sell_wide_kernel") and k.endswith(",1>"), k   # WALK = 1: groups
    assert torch.equal(res[False][0], res[True][0]) and res[False][1:] == res[True][1:]
    y = torch.empty_like(b)
    hipk.spmv(h, res[True][0], out=y)                                     # plain kernels (path still plain_only)
    true_res = float(torch.linalg.vector_norm(b - y))
    assert abs(true_res - res[True][3]) <= 1e-9 * float(torch.linalg.vector_norm(b))
    h.set_path(plain_only=False)


@pytest.mark.parametrize("solver", ["pbicgstab", "pgmres", "bicgstab", "gmres"])
def test_two_rows_per_lane_inside_the_solvers(hipk, solver):
    """The solvers' SpMV forms on the two-rows-per-lane kernel at a size that takes it (N = 2.25 M): a few iterations of
    BiCGStab / GMRES, plain and Jacobi-preconditioned (the epilogue's row scaling), identical to the plain CSR kernels."""
    from pytorch_sparse_solver.utils.matrix_utils import create_convdiff_2d_csr
    A = create_convdiff_2d_csr(1500, 1500, device=DEV)
    h = hipk.handle_for(A)
    n = A.shape[0]
    assert h.path() == "coded"
    b = torch.randn(n, dtype=torch.float64, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1))
    dinv = 1.0 / torch.linspace(3.0, 5.0, n, dtype=torch.float64, device=DEV)
    res = {}
    for plain in (False, True):
        h.set_path(plain_only=plain)
        x = torch.zeros_like(b)
        if solver == "pbicgstab":
            st = hipk.solve_pcg(h, dinv, b, x, tol=1e-10, atol=0.0, maxiter=12, method="bicgstab")
        elif solver == "pgmres":
            st = hipk.solve_pgmres(h, dinv, b, x, tol=1e-10, atol=0.0, maxiter=2, restart=6)
        else:
            st = hipk.solve(solver, h, b, x, tol=1e-10, atol=0.0, maxiter=12 if solver == "bicgstab" else 2, restart=6)
        res[plain] = (x.cpu().numpy(), st.iterations, st.info, st.residual_norm)
        if not plain:
            assert hipk.CsrHandle.last_spmv_kernel().startswith("hipk_spmv_sell_wide_kernel"), hipk.CsrHandle.last_spmv_kernel()
    h.set_path(plain_only=False)
    assert np.array_equal(res[False][0], res[True][0]) and res[False][1:] == res[True][1:]


# ------------------------------------------------------------------ offset-coded layout (variable coefficients)
@pytest.mark.parametrize("chunked", ["1", "0"])
@pytest.mark.parametrize("offsets", [[-300, -1, 0, 1, 300], [-1, 0, 1], [-40, -7, -1, 0, 1, 7, 40], [0, 5, 9, 11, 50, 51],
                                     list(range(-13, 14))])
@pytest.mark.parametrize("n", [1, 255, 257, 1023, 70_001, 1_200_011])
def test_offset_coded_random_values_all_widths(hipk, oracle, n, offsets, chunked, monkeypatch):
    monkeypatch.setenv("HIPK_SPMV_SELL_CHUNKED", chunked)
    rng = np.random.default_rng(n + len(offsets))
    crow, col, val = banded(n, offsets, lambda r, k: rng.standard_normal(len(r)))
    h = make_handle(hipk, crow, col, val, n)
    x = rng.standard_normal(n)
    if n == 1 and 0 not in offsets:
        return                                                                # empty matrix
    expect = "coded" if len(val) <= 255 else "offset_coded"                  # every value distinct: pairs = entries
    y_coded, y_plain = both_paths(hipk, h, torch.from_numpy(x).to(DEV), expect=expect)
    ref = oracle.spmv(crow, col, val, x)
    assert np.array_equal(y_coded, ref) and np.array_equal(y_plain, ref)
    assert n < 1000 or h.format_bytes() < h.spmv_bytes()                     # tiny systems are all tile padding


@pytest.mark.parametrize("groups", ["0", "1"])
def test_offset_coded_fused_dots_fp32_and_whole_solves(hipk, oracle, groups, monkeypatch):
    """groups = 1: the one-row-per-lane chunk kernel on a grid of groups of 4 tiles (what it takes at N > 16 M and on a rank's
    row block: hipk_spmv_args::group_tiles), forced at this size; value planes, fp64 and fp32 storage."""
    from pytorch_sparse_solver.utils.matrix_utils import create_variable_diffusion_2d_csr
    monkeypatch.setenv("HIPK_SPMV_SELL_STRIDED", groups)
    A = create_variable_diffusion_2d_csr(1500, 1500, device=DEV)             # 2.25M rows: chunked persistent kernel
    n = A.shape[0]
    h = hipk.handle_for(A)
    assert h.path() == "offset_coded" and h.format_bytes() < 0.8 * h.spmv_bytes()
    g = torch.Generator(device=DEV).manual_seed(7)
    x, w, b = (torch.randn(n, dtype=torch.float64, device=DEV, generator=g) for _ in range(3))
    coded = _spmv_ex_all_modes(hipk, h, x, w, b)
    assert hipk.CsrHandle.last_spmv_kernel().endswith("/groups") == (groups == "1"), hipk.CsrHandle.last_spmv_kernel()
    res = {}
    for plain in (False, True):
        h.set_path(plain_only=plain)
        xs = torch.zeros_like(b)
        st = hipk.solve("cg", h, b, xs, tol=1e-10, atol=0.0, maxiter=150)
        res[plain] = (xs.cpu().numpy(), st.iterations, st.residual_norm)
    plain = _spmv_ex_all_modes(hipk, h, x, w, b)
    h.set_path(plain_only=False)
    for a, c in zip(coded, plain):
        assert np.array_equal(a, c)
    assert np.array_equal(res[False][0], res[True][0]) and res[False][1:] == res[True][1:]
    crow, col, val = (t.cpu().numpy() for t in (A.crow_indices(), A.col_indices(), A.values()))
    assert np.array_equal(coded[0], b.cpu().numpy() - oracle.spmv(crow, col, val, x.cpu().numpy()))
    A32 = torch.sparse_csr_tensor(A.crow_indices(), A.col_indices(), A.values().float(), size=A.shape)
    h32 = hipk.handle_for(A32)
    assert h32.path() == "offset_coded"
    x32 = x.float()
    y32, y32p = both_paths(hipk, h32, x32, expect="offset_coded")
    assert np.array_equal(y32, y32p) and np.array_equal(y32, oracle.spmv32(crow, col, val.astype(np.float32), x32.cpu().numpy()))


@pytest.mark.parametrize("groups", ["0", "1"])
def test_fp32_storage_poisson_chunk_and_group_walks(hipk, groups, monkeypatch):
    """fp32 storage of the headline matrix's kind (1500 x 1500 Poisson: pair codes, uniform tiles, but not the fp64-only
    two-rows-per-lane kernel): the one-row-per-lane kernels with a workgroup per chunk and, forced, per group of 4 tiles; 30 CG
    iterations and all fused-dot modes bitwise equal to the general CSR kernels."""
    from pytorch_sparse_solver.utils.matrix_utils import create_poisson_2d_csr
    monkeypatch.setenv("HIPK_SPMV_SELL_STRIDED", groups)
    A = create_poisson_2d_csr(1500, 1500, device=DEV)
    A32 = torch.sparse_csr_tensor(A.crow_indices(), A.col_indices(), A.values().float(), size=A.shape)
    n = A.shape[0]
    h = hipk.handle_for(A32)
    assert h.path() == "coded"
    g = torch.Generator(device=DEV).manual_seed(9)
    x, w, b = (torch.randn(n, dtype=torch.float32, device=DEV, generator=g) for _ in range(3))
    coded = _spmv_ex_all_modes(hipk, h, x, w, b)
    k = hipk.CsrHandle.last_spmv_kernel()
    assert k.startswith("hipk_spmv_sell_") and k.endswith("/groups") == (groups == "1"), k
    res = {}
    for plain in (False, True):
        h.set_path(plain_only=plain)
        xs = torch.zeros_like(b)
        st = hipk.solve("cg", h, b, xs, tol=1e-6, atol=0.0, maxiter=30)
        res[plain] = (xs.cpu().numpy(), st.iterations, st.residual_norm)
    plain = _spmv_ex_all_modes(hipk, h, x, w, b)
    h.set_path(plain_only=False)
    for a, c in zip(coded, plain):
        assert np.array_equal(a, c)
    assert np.array_equal(res[False][0], res[True][0]) and res[False][1:] == res[True][1:]


# ------------------------------------------------------------------ randomized structures through every path
@pytest.mark.parametrize("seed", range(40))
def test_fuzz_structures_all_paths_agree_with_the_oracle(hipk, oracle, seed):
    """Random banded / ragged / mixed matrices: whatever path the structure analysis picks (coded, offset-coded, CSR-ordered
    codes, tile, tile-fast, row-per-wavefront) must give the oracle's bits for y and for the fused-dot chunk partials."""
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice([37, 256, 300, 1025, 2049, 5000, 20_011]))
    kind = seed % 5
    if kind == 0:      # stencil with few values, some rows emptied
        offs = np.unique(rng.integers(-40, 41, size=rng.integers(1, 9)))
        vals = rng.standard_normal(len(offs)).round(1) + 0.05
        crow, col, val = banded(n, offs, lambda r, k: vals[k])
    elif kind == 1:    # stencil with random values (offset-coded)
        offs = np.unique(rng.integers(-300, 301, size=rng.integers(2, 12)))
        crow, col, val = banded(n, offs, lambda r, k: rng.standard_normal(len(r)))
    elif kind == 2:    # ragged rows, few offsets relative to the row (CSR-ordered codes or tile)
        lens = rng.integers(0, 33, n) * (rng.random(n) < 0.3)
        crow = np.concatenate([[0], np.cumsum(lens)])
        col = np.concatenate([np.sort((r + np.arange(l) * 3) % n) for r, l in enumerate(lens)]) if crow[-1] else np.zeros(0, np.int64)
        val = np.ones(int(crow[-1])) * 0.5
    elif kind == 3:    # random columns, short rows (plain tile kernels)
        lens = rng.integers(0, 12, n)
        crow = np.concatenate([[0], np.cumsum(lens)])
        col = np.concatenate([np.sort(rng.choice(n, size=l, replace=False)) for l in lens]) if crow[-1] else np.zeros(0, np.int64)
        val = rng.standard_normal(int(crow[-1]))
    else:              # a few long rows among short ones
        lens = np.where(rng.random(n) < 0.02, rng.integers(33, min(n, 600), n), rng.integers(1, 6, n))
        crow = np.concatenate([[0], np.cumsum(lens)])
        col = np.concatenate([np.sort(rng.choice(n, size=l, replace=False)) for l in lens])
        val = rng.standard_normal(int(crow[-1]))
    crow, col, val = np.asarray(crow, np.int64), np.asarray(col, np.int64), np.asarray(val, np.float64)
    h = make_handle(hipk, crow, col, val, n)
    g = torch.Generator(device=DEV).manual_seed(seed)
    x, w, b = (torch.randn(n, dtype=torch.float64, device=DEV, generator=g) for _ in range(3))
    auto = _spmv_ex_all_modes(hipk, h, x, w, b)
    h.set_path(plain_only=True)
    plain = _spmv_ex_all_modes(hipk, h, x, w, b)
    for a, c in zip(auto, plain):
        assert np.array_equal(a, c), h.path()
    y_ref = b.cpu().numpy() - oracle.spmv(crow, col, val, x.cpu().numpy())
    assert np.array_equal(auto[0], y_ref)
    ch = int(hipk.lib().hipk_chunk_size(n))
    G = int(hipk.lib().hipk_chunk_count(n))
    p0 = np.zeros(G)
    p1 = np.zeros(G)
    wv = w.cpu().numpy()
    for c in range(G):
        sl = slice(c * ch, min(n, (c + 1) * ch))
        p0[c] = oracle.dot_tiled(wv[sl], y_ref[sl])
        p1[c] = oracle.dot_tiled(y_ref[sl], y_ref[sl])
    assert np.array_equal(auto[1], p0) and np.array_equal(auto[2], p1)


@pytest.mark.parametrize("case", ["poisson", "convdiff", "vardiff", "ldc", "wide11"])
def test_uniform_tile_shortcut_on_and_off(hipk, oracle, case, monkeypatch):
    """Tiles whose 256 rows share their code bytes are served from one 8-byte word per tile (hipk_tile_uniform_kernel);
    HIPK_SPMV_UNIFORM=0 reads every tile's planes.  Same bits, fewer bytes streamed."""
    from pytorch_sparse_solver.utils import matrix_utils as mu
    if case == "wide11":                                                  # 11 entries per row: three code groups, never uniform
        n = 40_000
        crow, col, val = banded(n, [-500, -9, -4, -2, -1, 0, 1, 2, 4, 9, 500], lambda r, k: 1.0 + k)
    else:
        # grid lines much longer than a 256-row tile, so that most tiles hold no line end
        A = {"poisson": lambda: mu.create_poisson_2d_csr(41, 3001, device="cpu"),
             "convdiff": lambda: mu.create_convdiff_2d_csr(37, 3000, device="cpu"),
             "vardiff": lambda: mu.create_variable_diffusion_2d_csr(2100, 2100, device="cpu"),
             "ldc": lambda: mu.create_ldc_pressure_csr(1100, device="cpu")}[case]()
        n = A.shape[0]
        crow, col, val = (A.crow_indices().numpy().astype(np.int64), A.col_indices().numpy().astype(np.int64),
                          A.values().numpy())
    x = np.random.default_rng(11).standard_normal(n)
    ref = oracle.spmv(crow, col, val, x)
    fb = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("HIPK_SPMV_UNIFORM", flag)
        h = make_handle(hipk, crow, col, val, n)
        assert h.path() in ("coded", "offset_coded")
        y = hipk.spmv(h, torch.from_numpy(x).to(DEV)).cpu().numpy()
        assert np.array_equal(y, ref)
        fb[flag] = h.format_bytes()
    if case == "wide11":
        assert fb["1"] == fb["0"]
    else:
        assert fb["1"] < fb["0"]
