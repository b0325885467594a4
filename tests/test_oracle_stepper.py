"""Replay the reference's state-injection unit tests (src/tests/*.rs) against the CPU oracle.

Each reference test builds an `Ida<Dummy,..>`, overwrites private fields with a "before" snapshot, calls one
private method and compares with the "after" snapshot (17-digit dumps from instrumented C IDA). The literals live
in tests/golden/stepper_goldens.json (extracted by tools/extract_goldens.py, provenance inside).
Tolerances follow the reference: `assert_eq!` -> exact, `assert_nearly_eq!` -> nearly_eq default (we use 1e-14 rel,
far tighter than the crate's default), printed-16-digit goldens -> 1e-15 rel.
"""
import json
import os

import numpy as np

import oracle_lib as O

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "stepper_goldens.json")))


def before_after(bindings):
    before, after = {}, {}
    for k, v in bindings:
        k = k.split(".")[-1]
        k = k[4:] if k.startswith("ida_") else k
        (after if k in before else before)[k] = v
    return before, after


def dummy():
    return O.OracleIda("dummy", 3, [0.0, 0.0, 0.0], [0.0, 0.0, 0.0], 1e-4, 1e-4)


def close(a, b, rel=1e-14):
    a, b = np.asarray(a, dtype=float).ravel(), np.asarray(b, dtype=float).ravel()
    return np.all(np.abs(a - b) <= rel * np.maximum(np.abs(a), np.abs(b)))


def test_set_coeffs_goldens():
    for t in ("test1", "test2"):
        b, a = before_after(G["set_coeffs"][t]["bindings"])
        ida = dummy()
        for k in ("hh", "hused", "ns", "kused", "kk", "cj", "cjlast"):
            ida.set(k, b[k])
        for k in ("beta", "alpha", "gamma", "sigma", "phi", "psi"):
            ida.setv(k, b[k])
        ck = O.lib().oracle_ida_set_coeffs(ida.h)
        assert close(ck, b["ck_expect"]), t
        for k in ("ns", "kused", "kk"):
            assert ida.get(k) == a[k], (t, k)
        for k in ("hh", "hused", "cj", "cjlast"):
            assert close(ida.get(k), a[k]), (t, k)
        for k in ("beta", "alpha", "gamma", "sigma", "phi", "psi"):
            assert close(ida.getv(k), a[k]), (t, k)
    # test1's ck literal has 19 digits: it is reproduced to the last bit
    b, _ = before_after(G["set_coeffs"]["test1"]["bindings"])
    ida = dummy()
    for k in ("hh", "hused", "ns", "kused", "kk", "cj", "cjlast"):
        ida.set(k, b[k])
    for k in ("beta", "alpha", "gamma", "sigma", "phi", "psi"):
        ida.setv(k, b[k])
    assert O.lib().oracle_ida_set_coeffs(ida.h) == b["ck_expect"]


def test_predict_golden():
    b, a = before_after(G["predict"]["test1"]["bindings"])
    ida = dummy()
    ida.set("kk", b["kk"])
    for k in ("phi", "gamma", "yypredict", "yppredict"):
        ida.setv(k, b[k])
    O.lib().oracle_ida_predict(ida.h)
    assert close(ida.getv("phi"), a["phi"])
    assert close(ida.getv("yypredict"), a["yypredict"])
    assert close(ida.getv("yppredict"), a["yppredict"], rel=1e-13)  # sums with cancellation, 17-digit literals


def test_restore_golden():
    bl = dict((k.split(".")[-1].replace("ida_", ""), v) for k, v in G["restore"]["test_restore1"]["bindings"])
    ida = dummy()
    for k in ("tn", "ns", "kk", "hh"):
        ida.set(k, bl[k])
    for k in ("phi", "psi", "cvals", "beta"):
        ida.setv(k, bl[k])
    O.lib().oracle_ida_restore(ida.h, bl["saved_t"])
    assert ida.get("tn") == bl["saved_t"]
    assert ida.get("ns") == 1 and ida.get("kk") == 2
    assert close(ida.getv("cvals"), bl["cvals_after"])
    assert close(ida.getv("beta"), bl["beta_after"])
    assert close(ida.getv("psi"), bl["psi_after"])
    assert close(ida.getv("phi"), bl["phi_after"])


def test_get_solution_golden():
    b = dict(G["get_solution"]["test_get_solution"]["bindings"])
    ida = dummy()
    for k in ("hh", "tn", "kused", "hused"):
        ida.set(k, b[k])
    ida.setv("phi", b["ida_phi"])
    ida.setv("psi", b["ida_psi"])
    assert O.lib().oracle_ida_get_solution(ida.h, b["t"]) == 0
    assert close(ida.getv("yy"), b["yret_expect"], rel=1e-15)
    assert close(ida.getv("yp"), b["ypret_expect"], rel=1e-14)


def test_get_solution_rejects_t_before_last_step():
    b = dict(G["get_solution"]["test_get_solution"]["bindings"])
    ida = dummy()
    for k in ("hh", "tn", "kused", "hused"):
        ida.set(k, b[k])
    ida.setv("phi", b["ida_phi"])
    ida.setv("psi", b["ida_psi"])
    assert O.lib().oracle_ida_get_solution(ida.h, b["tn"] - 2.0 * b["hused"]) == -26  # IdaError::BadTimeValue


def test_test_error_goldens():
    for t, expect_pass in (("test1", False), ("test2", True)):
        b = dict(G["test_error"][t]["bindings"])
        ida = dummy()
        ida.set("kk", b["kk"])
        ida.set("suppressalg", b["suppressalg"])
        ida.setv("phi", b["ida_phi"])
        ida.setv("ee", b["ida_ee"])
        ida.setv("ewt", b["ida_ewt"])
        ida.setv("sigma", b["ida_sigma"])
        ek, ekm1 = O.C.c_double(), O.C.c_double()
        ok = O.lib().oracle_ida_test_error(ida.h, b["ck"], O.C.byref(ek), O.C.byref(ekm1))
        assert bool(ok) == expect_pass, t
        assert ida.get("knew") == b["knew"]
        assert close(ek.value, b["err_k"], rel=2e-15), (t, ek.value, b["err_k"])  # 16 printed digits
        assert close(ekm1.value, b["err_km1"], rel=2e-15), (t, ekm1.value, b["err_km1"])


def test_complete_step_goldens_exact():
    # the reference uses exact assert_eq! here (complete_step.rs:94-106, 188-200, 294-306), incl. powf results
    for t in ("test1", "test2", "test3"):
        b, a = before_after(G["complete_step"][t]["bindings"])
        ida = dummy()
        for k in ("nst", "kk", "hh", "rr", "kused", "hused", "knew", "maxord", "phase", "hmax_inv"):
            ida.set(k, b[k])
        for k in ("ee", "phi", "ewt"):
            ida.setv(k, b[k])
        O.lib().oracle_ida_complete_step(ida.h, b["err_k"], b["err_km1"])
        for k in ("nst", "kk", "hh", "rr", "kused", "hused", "knew", "maxord", "phase", "hmax_inv"):
            assert ida.get(k) == a[k], (t, k, ida.get(k), a[k])
        for k in ("ee", "ewt", "phi"):
            assert np.array_equal(ida.getv(k), np.asarray(a[k], dtype=float).ravel()), (t, k)


def test_nonlinear_solve_snapshot_tolerance_level():
    """src/tests/nonlinear_solve.rs is #[ignore]d upstream: the snapshot lacks the factored Jacobian and the
    Newton solver's `jcur`. Replaying it with a Jacobian freshly evaluated at the snapshot's (yy, yp, cj) --
    instead of the stale one the C run used -- reproduces the expected correction to Newton-tolerance level."""
    b, a = before_after(G["nonlinear_solve"]["test1"]["bindings"])
    ida = O.OracleIda("roberts", 3, [0.0, 0.0, 0.0], [0.0, 0.0, 0.0], 1e-4, 1e-4)
    for k in ("nst", "cjold", "cj", "ss", "cjratio", "eps_newt"):
        ida.set(k, b[k])
    ida.set("cjlast", b["cj"])
    ida.set("toldel", 0.0001 * b["eps_newt"])
    for k in ("delta", "ee", "ewt", "yy", "yp", "yypredict", "yppredict"):
        ida.setv(k, b[k])
    assert O.lib().oracle_ida_lsetup(ida.h) == 0
    ida.set("ss", b["ss"])  # lsetup resets ss to 20; the snapshot's ss is that of a stale-Jacobian step
    assert O.lib().oracle_ida_nonlinear_solve(ida.h) == 0
    ewt = np.array(a["ewt"])
    err = O.wrms(ida.getv("ee") - np.array(a["ee"]), ewt)
    assert err < 0.33 * 0.1, err  # within a tenth of eps_newt in the WRMS norm
    assert O.wrms(ida.getv("yy") - np.array(a["yy"]), ewt) < 0.33 * 0.1
    assert np.array_equal(ida.getv("yypredict"), np.array(a["yypredict"]))
    assert np.array_equal(ida.getv("delta"), np.zeros(3))
