"""End-to-end pin of the CPU oracle on config 1 (Roberts, examples/roberts.rs) and the Newton known-answer test
(crates/nonlinear/src/newton.rs:306-343).

Reference-side facts used (tests/golden/roberts_example.json, newton_golden.json):
  * the reference solution at t = 4e10 and the `check_ans` pass criterion (examples/roberts.rs:9-51),
  * 377 step attempts (frame count of the author's Rust-vs-C trace, scripts/data_trace.ipynb),
  * first step size h0 = 2.1649552860480770e-05 (the `hh` literal of src/tests/complete_step.rs:40).
SURVEY.md Appendix A lists the full table this run must regenerate (nst/k/h at every output, counters, roots).
"""
import json
import os

import numpy as np

import oracle_lib as O

GD = os.path.join(os.path.dirname(__file__), "golden")
R = json.load(open(os.path.join(GD, "roberts_example.json")))


def run_roberts():
    ida = O.OracleIda("roberts", 3, R["yy0"], R["yp0"], R["rtol"], R["atol"])
    rows, roots = [], []
    tout, iout = R["tout0"], 0
    while iout < R["nout"]:
        st, tret = ida.solve(tout)
        assert st >= 0, st
        rows.append((tret, ida.getv("yy"), int(ida.get("nst")), int(ida.get("kused")), ida.get("hused"), st))
        if st == 2:
            roots.append((tret, ida.getv("iroots")))
        elif st == 0:
            iout += 1
            tout *= R["tout_factor"]
    return ida, rows, roots


def test_newton_known_answer():
    g = dict(json.load(open(os.path.join(GD, "newton_golden.json")))["test_newton"]["bindings"])
    y0, w = O.f64(g["y0"]), O.f64(g["w"])
    y = np.zeros(3)
    ni, nf = O.C.c_long(), O.C.c_long()
    r = O.lib().oracle_newton_test(O._ptr(y0), O._ptr(w), 1e-2, 10, O._ptr(y), O.C.byref(ni), O.C.byref(nf))
    assert r == 0
    err = y - np.array(g["y_exp"])
    exp = np.array(g["expected_err"])
    assert np.all(np.abs(err - exp) <= 1e-5 * np.maximum(np.abs(err), np.abs(exp)))  # assert_relative_eq max_relative=1e-5
    assert nf.value == 0


def test_roberts_counters_and_answer():
    ida, rows, roots = run_roberts()
    c = ida.counters()
    # SURVEY.md Appendix A (consistent with the reference's recorded 377 trace frames)
    assert c["n_attempts"] == R["trace_frames"]["step_attempts"] == 377
    assert (c["nst"], c["nre"], c["nje"], c["nsetups"], c["nni"], c["netf"], c["ncfn"]) == (362, 537, 60, 60, 537, 15, 0)
    assert c["nge"] == 404
    assert c["nls_nconvfails"] == 5
    # check_ans (examples/roberts.rs:9-51)
    y = rows[-1][1]
    ref = np.array(R["reference_solution_t4e10"])
    ewt = 1.0 / (R["rtol"] * np.abs(ref) + 10.0 * np.array(R["atol"]))
    assert O.wrms(y - ref, ewt) < 1.0
    assert abs(O.wrms(y - ref, ewt) - 0.01987) < 1e-4
    # bits of y(4e10) recorded in SURVEY.md Appendix A
    assert [v.hex() for v in y] == ["0x1.a1d277a766cb0p-25", "0x1.b61e4814ea4bbp-43", "0x1.fffffe5e2d1adp-1"]


def test_roberts_first_step_size():
    ida = O.OracleIda("roberts", 3, R["yy0"], R["yp0"], R["rtol"], R["atol"])
    ida.solve(0.4)
    assert ida.get("h0u") == 2.1649552860480770e-05


def test_roberts_output_table():
    _, rows, roots = run_roberts()
    # (t, nst, k) at every return, SURVEY.md Appendix A
    expect = [(2.64016e-01, 27, 2), (4.0e-01, 29, 3), (4.0e+00, 43, 4), (4.0e+01, 68, 4), (4.0e+02, 95, 4), (4.0e+03, 126, 3),
              (4.0e+04, 161, 5), (4.0e+05, 202, 3), (4.0e+06, 250, 3), (2.07880e+07, 280, 5), (4.0e+07, 293, 4),
              (4.0e+08, 325, 4), (4.0e+09, 348, 3), (4.0e+10, 362, 2)]
    assert len(rows) == len(expect)
    for (t, y, nst, k, h, st), (te, nste, ke) in zip(rows, expect):
        assert abs(t - te) <= 1e-5 * te and nst == nste and k == ke
    assert len(roots) == 2
    assert roots[0][1].tolist() == [0.0, -1.0] and roots[1][1].tolist() == [1.0, 0.0]  # Q8: signum(glo)
    hs = [r[4] for r in rows]
    assert abs(hs[1] - 8.80238e-02) < 1e-6 and abs(hs[-1] - 7.54805e+09) < 1e5
