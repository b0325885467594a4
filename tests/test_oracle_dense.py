"""Pin the CPU oracle's LU / solve / WRMS against the reference's own unit goldens
(crates/linear/src/dense.rs:208-329, src/norm_rms.rs:64-86) -- bit-exact where the reference asserts
`assert_eq!`, 1e-9 relative where it asserts `assert_relative_eq!`."""
import json
import math
import os

import numpy as np

import oracle_lib as O

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "dense_goldens.json")))


def _b(name):
    return dict((k, v) for k, v in G[name]["bindings"])


def test_get_rf_goldens_bit_exact():
    for name in ("test_get_rf1", "test_get_rf2"):
        b = _b(name)
        # the second `pivot` binding does not exist: the expected pivots are the literal [2,1,2] in the assert
        info, lu, piv = O.getrf(np.array(b["mat_a"]))
        assert info == 0
        assert np.array_equal(lu, np.array(b["expect"])), name  # exact bits (assert_eq! in the reference)
        assert piv.tolist() == [2, 1, 2]


def test_get_rs_goldens_bit_exact():
    for name in ("test_get_rs1", "test_get_rs2"):
        b = _b(name)
        x = O.getrs(np.array(b["mat_a"]), np.array(b["pivot"], dtype=np.int64), np.array(b["b"]))
        assert np.array_equal(x, np.array(b["expect"])), name


def test_dense1_lsolver_trait():
    b = _b("test_dense1")
    a = np.asfortranarray(np.array(b["mat_a"])).copy(order="F")
    rhs = O.f64(b["b"])
    x = np.zeros(4)
    piv = np.zeros(4, dtype=np.int64)
    info = O.lib().oracle_dense_lsolver(a.ctypes.data_as(O.dp), 4, O._ptr(rhs), O._ptr(x), O._ptr(piv, O.i64p))
    assert info == 0
    assert np.allclose(x, b["expected"], rtol=1e-9, atol=0)


def test_zero_pivot_reports_one_based_column():
    a = np.array([[1.0, 2.0, 3.0], [2.0, 4.0, 6.0], [1.0, 1.0, 1.0]])  # rank 2: column 2 has no pivot after step 1? no -> col 3
    info, _, _ = O.getrf(a)
    assert info in (2, 3) and info > 0
    z = np.zeros((3, 3))
    assert O.getrf(z)[0] == 1


def test_pivot_tie_keeps_lowest_row():
    a = np.array([[1.0, 0.5], [-1.0, 2.0]])  # |a00| == |a10| -> strict '>' keeps row 0
    info, _, piv = O.getrf(a)
    assert info == 0 and piv[0] == 0


def test_wrms_golden():
    w = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "wrms_golden.json")))
    n = w["length"]
    x = np.full(n, w["x"])
    ww = np.full(n, w["w"])
    assert O.wrms(x, ww) == w["expect"]  # exact (assert_eq! in the reference)
    idm = np.ones(n, dtype=np.uint8)
    idm[w["masked"]["masked_out_index"]] = 0
    fac = math.sqrt((n - 1) / n)
    got = O.lib().oracle_norm_wrms_masked(O._ptr(x), O._ptr(ww), idm.ctypes.data_as(O.C.POINTER(O.C.c_uint8)), n)
    assert got == fac * 0.5 * 0.5


def test_lu_solve_random_against_numpy():
    rng = np.random.default_rng(0)
    for n in (1, 2, 5, 17, 64):
        a = rng.standard_normal((n, n))
        xs = rng.standard_normal(n)
        info, lu, piv = O.getrf(a)
        assert info == 0
        x = O.getrs(lu, piv, a @ xs)
        assert np.allclose(x, xs, rtol=1e-8, atol=1e-10)
