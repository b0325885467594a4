"""GPU parity of the LSolver entry points (idahip_ls_setup / idahip_ls_solve / idahip_wrms) against the CPU oracle
and the reference's own goldens. Bar: bit-exact LU factors, pivots, solutions and norms."""
import json
import math
import os

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
GD = os.path.join(os.path.dirname(__file__), "golden")


def colmajor(mats):
    """[B][n][n] logical -> contiguous column-major storage per system."""
    return np.ascontiguousarray(np.transpose(mats, (0, 2, 1)))


LU_VARIANT = 4


@pytest.fixture(params=[3, 4], ids=["lu-panel2", "lu-wavepanel"], autouse=True)
def lu_variant(request):
    """Every test of this file runs against every factorisation pipeline."""
    global LU_VARIANT
    LU_VARIANT = request.param
    yield


def gpu_lu(mats, idx=None):
    import idahip
    B, n, _ = mats.shape
    ctx = idahip.Ctx("linear_dense" if n != 3 else "lorenz63", n, B)
    ctx.set_lu_variant(LU_VARIANT)
    dA = ctx.dev_array(colmajor(mats))
    dP = ctx.dev_empty(8 * B * n)
    rc, info = ctx.ls_setup(dA, dP, idx)
    lu = np.transpose(ctx.to_host(dA, (B, n, n)), (0, 2, 1))
    piv = ctx.to_host(dP, (B, n), dtype=np.int64)
    return ctx, dA, dP, rc, info, lu, piv


def oracle_lu(mats):
    out = [O.getrf(m) for m in mats]
    return np.array([o[0] for o in out]), np.array([o[1] for o in out]), np.array([o[2] for o in out])


def test_reference_lu_goldens_on_gpu():
    G = json.load(open(os.path.join(GD, "dense_goldens.json")))
    mats = np.array([dict(G[k]["bindings"])["mat_a"] for k in ("test_get_rf1", "test_get_rf2")])
    exp = np.array([dict(G[k]["bindings"])["expect"] for k in ("test_get_rf1", "test_get_rf2")])
    ctx, dA, dP, rc, info, lu, piv = gpu_lu(mats)
    assert rc == 0 and not info.any()
    assert np.array_equal(lu, exp)  # assert_eq! in the reference (dense.rs:287,310)
    assert piv.tolist() == [[2, 1, 2], [2, 1, 2]]


def test_reference_solve_goldens_on_gpu():
    import idahip
    G = json.load(open(os.path.join(GD, "dense_goldens.json")))
    names = ("test_get_rs1", "test_get_rs2")
    lus = np.array([dict(G[k]["bindings"])["mat_a"] for k in names])
    bs = np.array([dict(G[k]["bindings"])["b"] for k in names])
    pv = np.array([dict(G[k]["bindings"])["pivot"] for k in names], dtype=np.int64)
    exp = np.array([dict(G[k]["bindings"])["expect"] for k in names])
    ctx = idahip.Ctx("lorenz63", 3, 2)
    dLU, dP, dB = ctx.dev_array(colmajor(lus)), ctx.dev_array(pv), ctx.dev_array(bs)
    dX = ctx.dev_empty(8 * 2 * 3)
    ctx.ls_solve(dLU, dP, dX, dB)
    assert np.array_equal(ctx.to_host(dX, (2, 3)), exp)  # exact bits (dense.rs:238,264)


@pytest.mark.parametrize("n", [2, 3, 8, 9, 31, 32, 33, 64, 65, 100, 127, 257, 512])
def test_lu_and_solve_random_bit_exact(n):
    rng = np.random.default_rng(1000 + n)
    B = 5 if n < 200 else 3
    mats = rng.standard_normal((B, n, n))
    if n >= 9:
        mats[1, :, 3] = 0.0          # exact zeros in U rows -> exercises the a_kj == 0 skip paths
        mats[1, 3, 3] = 2.0
        mats[2][np.abs(mats[2]) < 0.8] = 0.0
        mats[2] += np.diag(np.full(n, 4.0))
    rhs = rng.standard_normal((B, n))
    info_o, lu_o, piv_o = oracle_lu(mats)
    ctx, dA, dP, rc, info, lu, piv = gpu_lu(mats)
    assert np.array_equal(info, info_o) and rc == 0
    assert np.array_equal(piv, piv_o)
    assert np.array_equal(lu, lu_o)
    dB = ctx.dev_array(rhs)
    dX = ctx.dev_empty(rhs.nbytes)
    ctx.ls_solve(dA, dP, dX, dB)
    x = ctx.to_host(dX, (B, n))
    x_o = np.array([O.getrs(lu_o[s], piv_o[s], rhs[s]) for s in range(B)])
    assert np.array_equal(x, x_o)


@pytest.mark.parametrize("n", [128, 129, 191, 192, 193, 320, 449, 511, 513, 640, 1000, 1024])
def test_lu_sizes_around_the_super_panel_and_slot_boundaries(n):
    """64-column super-panels, 64-row register slots of the wave-per-matrix panel kernel (<= 512 live rows), the two-rows-per-lane
    panels above that: sizes on both sides of every boundary, one dense matrix and one with many exact zeros."""
    rng = np.random.default_rng(5000 + n)
    mats = rng.standard_normal((2, n, n))
    mats[1][np.abs(mats[1]) < 0.9] = 0.0
    mats[1] += np.diag(np.full(n, 4.0))
    info_o, lu_o, piv_o = oracle_lu(mats)
    ctx, dA, dP, rc, info, lu, piv = gpu_lu(mats)
    assert rc == 0 and np.array_equal(info, info_o)
    assert np.array_equal(piv, piv_o)
    assert np.array_equal(lu, lu_o)


@pytest.mark.parametrize("n", [24, 100, 200, 600, 1100])
def test_nan_and_infinity_follow_the_reference_scan(n):
    """dense.rs:111-117 picks the pivot with `>` on absolute values: a NaN never wins a comparison, so it becomes the pivot only
    when it already sits at position k, and an infinity wins like any large value (and then breeds NaNs). Pivots and the
    zero-pivot verdict must match exactly; values bit for bit where they are numbers, NaN where the oracle has NaN (sign and
    payload of a NaN are not part of the contract: x86 and the GPU generate different default NaNs)."""
    rng = np.random.default_rng(77 + n)
    mats = rng.standard_normal((6, n, n))
    mats[0, 5, 5] = np.nan                      # NaN on the diagonal: the scan keeps it once column 5 is reached (if still there)
    mats[1, n - 2, 3] = np.nan                  # NaN below the diagonal: never chosen in column 3, poisons its row
    mats[2, 7, 2] = np.inf                      # +inf wins column 2
    mats[3, n // 2, 0] = -np.inf                # -inf wins column 0
    mats[4, 1, 1] = np.nan
    mats[4, 9, 4] = np.inf
    mats[4, 11, 4] = -np.inf                    # two infinities in one column: the first in scan order wins
    mats[5, :, :] = np.where(rng.random((n, n)) < 0.02, np.nan, mats[5])
    info_o, lu_o, piv_o = oracle_lu(mats)
    ctx, dA, dP, rc, info, lu, piv = gpu_lu(mats)
    assert np.array_equal(info, info_o)
    ok = info_o == 0
    assert np.array_equal(piv[ok], piv_o[ok])
    assert np.array_equal(np.isnan(lu[ok]), np.isnan(lu_o[ok]))
    assert np.array_equal(lu[ok], lu_o[ok], equal_nan=True)


@pytest.fixture(params=[1, 0], ids=["superpanel", "panel-by-panel"])
def large_n_pipeline(request, monkeypatch):
    """n > 1024: a 64-column super-panel as ONE launch of lu_superpanel_kernel (default, round 5) or as round 4's eight 8-column
    panel launches with a narrow update after each (IDAHIP_LU_SUPERPANEL=0, read when a context is created): both against the
    oracle, hence against each other."""
    monkeypatch.setenv("IDAHIP_LU_SUPERPANEL", str(request.param))
    return request.param


@pytest.mark.parametrize("n", [1025, 1100, 1600, 2048])
def test_lu_beyond_1024_rows(n, large_n_pipeline):
    """More than 1024 rows: the super-panels are factored by the workgroup-per-matrix kernels (whatever the LU variant) -- whole, or
    in eight 8-column panels with eight rows per lane."""
    rng = np.random.default_rng(n)
    B = 2
    mats = rng.standard_normal((B, n, n))
    mats[1][np.abs(mats[1]) < 1.0] = 0.0           # many exact zeros: the a_kj == 0 paths
    mats[1] += np.diag(np.full(n, 5.0))
    rhs = rng.standard_normal((B, n))
    info_o, lu_o, piv_o = oracle_lu(mats)
    ctx, dA, dP, rc, info, lu, piv = gpu_lu(mats)
    assert rc == 0 and np.array_equal(info, info_o)
    assert np.array_equal(piv, piv_o)
    assert np.array_equal(lu, lu_o)
    dB = ctx.dev_array(rhs)
    dX = ctx.dev_empty(rhs.nbytes)
    ctx.ls_solve(dA, dP, dX, dB)
    x_o = np.array([O.getrs(lu_o[s], piv_o[s], rhs[s]) for s in range(B)])
    assert np.array_equal(ctx.to_host(dX, (B, n)), x_o)


@pytest.mark.parametrize("n", [1536, 2120])
def test_banded_and_partly_banded_matrices_beyond_1024_rows(n, large_n_pipeline):
    """The large-n trailing updates treat nearly empty U12 blocks apart (zero column blocks found by their own kernel, the
    rest applied one live row per thread, helpers sharing the rows once the first super-panel has shown a band): a
    tridiagonal matrix, a wider band that pivots, a band that turns dense after the first super-panel (helpers on the
    dense path) and a matrix with dense leading rows (no band seen) all factor like the reference."""
    rng = np.random.default_rng(n + 7)
    B = 4
    m = np.zeros((B, n, n))
    i = np.arange(n)
    m[0, i, i] = 4.0 + rng.random(n)
    m[0, i[1:], i[:-1]] = -1.0 - rng.random(n - 1)
    m[0, i[:-1], i[1:]] = -1.0 - rng.random(n - 1)
    for d in range(-3, 4):  # band of 7, weak diagonal: partial pivoting swaps rows and widens U
        k = np.arange(max(0, -d), min(n, n - d))
        m[1, k, k + d] = rng.standard_normal(k.size) * (0.3 if d == 0 else 1.0)
    m[2] = m[0]
    m[2, 64:, 64:] = rng.standard_normal((n - 64, n - 64))  # banded where the band is measured, dense behind it
    m[3] = m[0]
    m[3, :40, :] = rng.standard_normal((40, n))             # dense leading rows: every column block has work
    rhs = rng.standard_normal((B, n))
    info_o, lu_o, piv_o = oracle_lu(m)
    ctx, dA, dP, rc, info, lu, piv = gpu_lu(m)
    assert rc == 0 and np.array_equal(info, info_o)
    assert np.array_equal(piv, piv_o)
    assert np.array_equal(lu, lu_o)
    dB = ctx.dev_array(rhs)
    dX = ctx.dev_empty(rhs.nbytes)
    ctx.ls_solve(dA, dP, dX, dB)
    x_o = np.array([O.getrs(lu_o[s], piv_o[s], rhs[s]) for s in range(B)])
    assert np.array_equal(ctx.to_host(dX, (B, n)), x_o)


def test_large_n_pipeline_defaults_and_setter(monkeypatch):
    """The structure hint of include/ida_hip.h: a super-panel per launch is the default for the heat problem (banded Jacobians),
    panel by panel for every other kind (dense matrices); idahip_set_lu_superpanel switches a context over, and the factors do not
    depend on it."""
    import idahip
    monkeypatch.delenv("IDAHIP_LU_SUPERPANEL", raising=False)
    h = idahip.Ctx("heat1d", 1100, 2)
    assert h.lu_superpanel() == 1
    h.close()
    n = 1100
    rng = np.random.default_rng(5)
    m = np.zeros((2, n, n))
    i = np.arange(n)
    for s in range(2):
        m[s, i, i] = 3.0 + rng.random(n)
        m[s, i[1:], i[:-1]] = -1.0 - rng.random(n - 1)
        m[s, i[:-1], i[1:]] = -1.0 - rng.random(n - 1)
    m[1] += np.where(rng.random((n, n)) < 0.01, rng.standard_normal((n, n)), 0.0)  # a band with some scattered entries
    res = []
    for on in (0, 1):
        ctx = idahip.Ctx("linear_dense", n, 2)
        assert ctx.lu_superpanel() == 0
        ctx.set_lu_superpanel(on)
        assert ctx.lu_superpanel() == on
        dA = ctx.dev_array(colmajor(m))
        dP = ctx.dev_empty(8 * 2 * n)
        rc, info = ctx.ls_setup(dA, dP, None)
        assert rc == 0 and not info.any()
        res.append((ctx.to_host(dA, (2, n, n)), ctx.to_host(dP, (2, n), dtype=np.int64)))
        ctx.close()
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    info_o, lu_o, piv_o = oracle_lu(m)
    assert np.array_equal(np.transpose(res[1][0], (0, 2, 1)), lu_o) and np.array_equal(res[1][1], piv_o)


def test_solve_through_the_buffer_loads_with_a_partial_last_block_at_large_n():
    """wg_getrs with 256 threads reads the diagonal blocks through a buffer descriptor with 32-bit byte counts and scalar
    offsets (n * n * 8 and (kb + k) * n * 8: exact up to n = 4096) and prefetches column groups past the end of a partial last
    block, whose lanes are all out of range. n = 2999: odd (one row per lane and load, the 256-thread kernel), 47 blocks, the
    last one 55 columns wide, offsets up to 72 MB."""
    n = 2999
    rng = np.random.default_rng(2999)
    mats = rng.standard_normal((1, n, n)) + np.eye(n) * 3.0
    rhs = rng.standard_normal((1, n))
    info_o, lu_o, piv_o = oracle_lu(mats)
    ctx, dA, dP, rc, info, lu, piv = gpu_lu(mats)
    assert rc == 0 and info_o[0] == 0 and np.array_equal(piv, piv_o) and np.array_equal(lu, lu_o)
    dB = ctx.dev_array(rhs)
    dX = ctx.dev_empty(rhs.nbytes)
    ctx.ls_solve(dA, dP, dX, dB)
    assert np.array_equal(ctx.to_host(dX, (1, n)), O.getrs(lu_o[0], piv_o[0], rhs[0])[None, :])


def test_nan_infinity_zero_pivot_and_ties_beyond_1024_rows(large_n_pipeline):
    """The special cases of the pivot scan (dense.rs:111-122) in the large-n kernels: NaN at and off the pivot position, two
    infinities in a column, exact ties in |a| between rows that different waves hold, an all-zero column (Err(k+1)), on matrices
    with many exact zeros (the a_kj == 0 rule in the U slot and in the left-looking updates of lu_superpanel_kernel)."""
    n = 1100
    rng = np.random.default_rng(11)
    m = rng.integers(-3, 4, size=(5, n, n)).astype(float)  # small integers: ties and zeros everywhere
    m += np.eye(n) * 2.0
    m[1, 700, 3] = np.nan          # NaN off the pivot position of its column
    m[1, 70, 70] = np.nan          # NaN at the pivot position
    m[2, 5, 64] = np.inf
    m[2, 900, 64] = -np.inf        # two infinities in one column: the first in scan order wins
    m[3, :, 130] = 0.0             # a zero column: zero pivot reported with its 1-based column, the other systems unaffected
    m[4] = np.where(rng.random((n, n)) < 0.001, np.nan, m[4])
    info_o, lu_o, piv_o = oracle_lu(m)
    ctx, dA, dP, rc, info, lu, piv = gpu_lu(m)
    assert np.array_equal(info, info_o) and info_o[3] != 0
    ok = info_o == 0
    assert np.array_equal(piv[ok], piv_o[ok])
    assert np.array_equal(lu[ok], lu_o[ok], equal_nan=True)


def test_ctx_factors_stay_exact_when_the_structure_changes_between_setups():
    """n >= 2048: the row scatter into the ctx's own factors does not write all-zero-bits values into 64 x 64 blocks that have only ever
    held zeros (lu_finalize_kernel, LuWs::dirty: the buffer starts as zeros and the map never resets). Successive setups on ONE ctx
    with changing structure -- a row-permuted band whose multipliers sit far from the diagonal, then a plain tridiagonal matrix
    (the far entries must be cleared again), then a dense one, then a band again -- each bit-identical to the oracle, the solve too."""
    import idahip
    if LU_VARIANT != 4:
        pytest.skip("one pipeline is enough for the scatter")
    n, B = 2048, 2
    rng = np.random.default_rng(20480)
    i = np.arange(n)

    def tri():
        m = np.zeros((n, n))
        m[i, i] = 4.0 + rng.random(n)
        m[i[1:], i[:-1]] = -1.0 - rng.random(n - 1)
        m[i[:-1], i[1:]] = -1.0 - rng.random(n - 1)
        return m

    def permuted_band():
        m = tri()
        p = np.arange(n)
        blk = rng.permutation(n // 64)           # whole 64-row groups trade places: pivoting brings them back, L reaches far
        p = (blk[:, None] * 64 + np.arange(64)[None, :]).ravel()
        return m[p]

    seq = [np.array([permuted_band(), tri()]), np.array([tri(), tri()]), rng.standard_normal((B, n, n)),
           np.array([tri(), permuted_band()])]
    ctx = idahip.Ctx("linear_dense", n, B)
    ctx.set_lu_variant(LU_VARIANT)
    zeros = np.zeros((B, n))
    cj = 3.25
    for step, Bm in enumerate(seq):
        A = np.array([np.eye(n) * 0.5 for _ in range(B)])
        ctx.set_linear_dense(colmajor(A), colmajor(Bm), zeros)
        rc, info = ctx.nls_lsetup(0.0, cj)
        assert rc == 0 and not info.any(), step
        for s_ in range(B):
            J = Bm[s_] + cj * A[s_]
            info_o, lu_o, piv_o = O.getrf(J)
            lu, piv = ctx.download_lu(s_)
            assert info_o == 0 and np.array_equal(piv, piv_o), (step, s_)
            assert np.array_equal(lu, lu_o), (step, s_)
            assert np.array_equal(np.signbit(lu), np.signbit(lu_o)), (step, s_)  # zeros with their signs: the map counts a -0.0 as content
    ctx.close()


def test_pivot_ties_resolve_like_the_reference_scan():
    n = 40
    rng = np.random.default_rng(7)
    m = rng.integers(-2, 3, size=(4, n, n)).astype(float)  # many exact ties in |a|, many zeros
    m += np.eye(n) * 3.0
    info_o, lu_o, piv_o = oracle_lu(m)
    ctx, dA, dP, rc, info, lu, piv = gpu_lu(m)
    assert np.array_equal(info, info_o)
    ok = info_o == 0
    assert np.array_equal(piv[ok], piv_o[ok]) and np.array_equal(lu[ok], lu_o[ok])


def test_banded_matrix_like_the_heat_jacobian():
    n = 96
    m = np.zeros((2, n, n))
    for s, coef in enumerate((3.0, 7.5)):
        for i in range(1, n - 1):
            m[s, i, i - 1] = -coef
            m[s, i, i] = 10.0 + 2 * coef
            m[s, i, i + 1] = -coef
        m[s, 0, 0] = m[s, n - 1, n - 1] = 1.0
    info_o, lu_o, piv_o = oracle_lu(m)
    ctx, dA, dP, rc, info, lu, piv = gpu_lu(m)
    assert not info.any() and np.array_equal(piv, piv_o) and np.array_equal(lu, lu_o)


def test_singular_matrix_reports_one_based_column_and_spares_the_others():
    n = 48
    rng = np.random.default_rng(3)
    m = rng.standard_normal((3, n, n))
    m[1, :, 10] = m[1, :, 4] * 2.0  # rank deficient -> exact zero pivot may or may not appear in fp; force an exact one:
    m[1] = 0.0
    m[1, np.arange(n), np.arange(n)] = 1.0
    m[1, 20, 20] = 0.0  # column 21 has no pivot
    info_o, lu_o, piv_o = oracle_lu(m)
    assert info_o[1] == 21
    ctx, dA, dP, rc, info, lu, piv = gpu_lu(m)
    assert rc == 1 and np.array_equal(info, info_o)
    for s in (0, 2):
        assert np.array_equal(lu[s], lu_o[s]) and np.array_equal(piv[s], piv_o[s])


def test_subset_of_systems_only():
    n = 70
    rng = np.random.default_rng(11)
    m = rng.standard_normal((6, n, n))
    info_o, lu_o, piv_o = oracle_lu(m)
    ctx, dA, dP, rc, info, lu, piv = gpu_lu(m, idx=[4, 1])
    assert np.array_equal(lu[4], lu_o[4]) and np.array_equal(lu[1], lu_o[1])
    for s in (0, 2, 3, 5):
        assert np.array_equal(lu[s], m[s])  # untouched


@pytest.mark.parametrize("n", [3, 32, 100, 512, 1000])
def test_wrms_matches_sequential_sum(n):
    import idahip
    rng = np.random.default_rng(n)
    B = 4
    x = rng.standard_normal((B, n)) * 10.0 ** rng.integers(-8, 8, size=(B, n))
    w = np.abs(rng.standard_normal((B, n))) + 0.1
    ctx = idahip.Ctx("linear_dense" if n != 3 else "lorenz63", n, B)
    got = ctx.wrms(ctx.dev_array(x), ctx.dev_array(w))
    exp = np.array([O.wrms(x[s], w[s]) for s in range(B)])
    assert np.array_equal(got, exp)


def test_wrms_reference_golden():
    import idahip
    g = json.load(open(os.path.join(GD, "wrms_golden.json")))
    n = g["length"]
    ctx = idahip.Ctx("linear_dense", n, 1)
    got = ctx.wrms(ctx.dev_array(np.full((1, n), g["x"])), ctx.dev_array(np.full((1, n), g["w"])))
    assert got[0] == g["expect"] == 0.25


def test_empty_and_invalid_system_lists():
    """nsys = 0 is a no-op for every entry point that takes a list; ids outside the batch or an unsupported size are
    refused with an error message before anything is launched (shapes are checked on the host: a bad index would be an
    out-of-bounds access on the device)."""
    import ctypes as C
    import idahip
    n, B = 24, 3
    rng = np.random.default_rng(0)
    mats = rng.standard_normal((B, n, n))
    ctx = idahip.Ctx("linear_dense", n, B)
    ctx.set_lu_variant(LU_VARIANT)
    dA = ctx.dev_array(colmajor(mats))
    dP = ctx.dev_empty(8 * B * n)
    before = ctx.to_host(dA, (B, n, n)).copy()
    empty = np.zeros(0, dtype=np.int32)
    rc, info = ctx.ls_setup(dA, dP, empty)
    assert rc == 0 and info.size == 0
    assert np.array_equal(ctx.to_host(dA, (B, n, n)), before)
    assert ctx.wrms(dA, dA, idx=empty).size == 0
    ctx.nls_sys(0.0, 1.0, True, idx=empty)
    rc, info = ctx.nls_lsetup(0.0, 1.0, idx=empty)
    assert rc == 0
    for bad in ([B], [-1], [0, 1, 7]):
        with pytest.raises(idahip.IdaHipError, match="out of range"):
            ctx.ls_setup(dA, dP, bad)
        with pytest.raises(idahip.IdaHipError, match="out of range"):
            ctx.nls_sys(0.0, 1.0, True, idx=bad)
    assert np.array_equal(ctx.to_host(dA, (B, n, n)), before)
    with pytest.raises(idahip.IdaHipError):
        idahip.Ctx("linear_dense", 5000, 1).ls_setup(ctx.dev_empty(8), ctx.dev_empty(8), [0])  # n > 4096: refused


def test_entry_points_run_on_the_ctx_device_whatever_device_is_current():
    """A ctx belongs to the device it was created on: every entry point switches to it and puts the caller's current
    device back (one host thread driving several GPUs). Needs two visible GPUs to see a switch; on a
    one-GPU box it still checks that the calls leave the current device alone."""
    import torch
    import idahip
    ndev = torch.cuda.device_count()
    other = 1 if ndev > 1 else 0
    n, B = 48, 3
    rng = np.random.default_rng(12)
    mats = rng.standard_normal((B, n, n))
    info_o, lu_o, piv_o = oracle_lu(mats)
    ctx = idahip.Ctx("linear_dense", n, B, device=0)
    torch.cuda.set_device(other)
    dA = ctx.dev_array(colmajor(mats))     # allocation, copies, launches: all with device `other` current in this thread
    dP = ctx.dev_empty(8 * B * n)
    rc, info = ctx.ls_setup(dA, dP, None)
    assert torch.cuda.current_device() == other
    lu = np.transpose(ctx.to_host(dA, (B, n, n)), (0, 2, 1))
    assert rc == 0 and np.array_equal(lu, lu_o) and np.array_equal(ctx.to_host(dP, (B, n), dtype=np.int64), piv_o)
    assert torch.cuda.current_device() == other
    torch.cuda.set_device(0)


def test_newton_solve_with_zero_blocks_in_the_factors_is_exact_in_the_corner_cases():
    """n >= 2048: the factorisation leaves a map of the factors' all-zero 64 x 64 blocks and the Newton iteration's triangular
    solves leave such blocks out -- but only where that is exactly what the reference's arithmetic gives: not when an entry
    of b they would multiply with is infinite or NaN, and not for a row whose entry is -0.0 (0 * b_k subtracted from -0.0 can
    flip its sign). J = B + cj A with block-sparse A, B; right-hand sides with -0.0, an infinity, and ordinary values."""
    import idahip
    n, B = 2048, 3
    rng = np.random.default_rng(20482)
    nb = n // 64
    Bm = np.zeros((B, n, n))
    for s in range(B):
        for q in range(nb):  # diagonal blocks, a sub- and a super-diagonal block here and there, one far block
            Bm[s, q * 64:(q + 1) * 64, q * 64:(q + 1) * 64] = rng.standard_normal((64, 64)) + 8.0 * np.eye(64)
            if q % 3 == 1:
                Bm[s, q * 64:(q + 1) * 64, (q - 1) * 64:q * 64] = rng.standard_normal((64, 64))
            if q % 5 == 2 and q + 1 < nb:
                Bm[s, q * 64:(q + 1) * 64, (q + 1) * 64:(q + 2) * 64] = rng.standard_normal((64, 64))
        Bm[s, 20 * 64:21 * 64, 3 * 64:4 * 64] = rng.standard_normal((64, 64))
    A = np.zeros((B, n, n))
    ctx = idahip.Ctx("linear_dense", n, B)
    ctx.set_tolerances(1e-6, 1e-8)
    ctx.set_linear_dense(colmajor(A), colmajor(Bm), np.zeros((B, n)))
    ctx.upload(idahip.F_YY, np.zeros((B, n)))
    ctx.upload(idahip.F_YP, np.zeros((B, n)))
    rc, info = ctx.nls_lsetup(0.0, 1.0)
    assert rc == 0 and not info.any()
    rhs = rng.standard_normal((B, n))
    rhs[0, 100:900] = 0.0        # the solve sees -0.0 there (the right-hand side is negated first)
    rhs[0, 1500:1600] = -0.0     # ... and +0.0 here
    rhs[1, 700] = np.inf         # spreads NaN / inf down the forward sweep and back up
    rhs[2, 5 * 64:9 * 64] = 0.0
    ctx.upload(idahip.F_DELTA, rhs)
    ctx.upload(idahip.F_EE, np.zeros((B, n)))
    ctx.upload(idahip.F_EWT, np.ones((B, n)))
    ctx.newton_iter(1.0)
    got = ctx.download(idahip.F_DELTA)
    for s in range(B):
        lu, piv = ctx.download_lu(s)
        want = O.getrs(lu, piv, -rhs[s])
        assert np.array_equal(got[s].view(np.uint64), want.view(np.uint64)) or (
            np.array_equal(np.isnan(got[s]), np.isnan(want)) and np.array_equal(got[s][~np.isnan(want)].view(np.uint64), want[~np.isnan(want)].view(np.uint64))), s
    ctx.close()
