"""include/ida_hip.h consumed from C: tests/native/c_consumer.c is compiled by gcc as C99 (-pedantic -Werror) against the two
headers and linked with libidahip.so, then plays LSolver::setup / LSolver::solve / NormRms::norm_wrms on the reference's own
3 x 3 goldens (crates/linear/src/dense.rs:216-311, src/norm_rms.rs:64-70) through the C ABI. The CPU half checks that it
compiles and links; the GPU half runs it and compares bit for bit."""
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GD = os.path.join(ROOT, "tests", "golden")
CSRC = os.path.join(ROOT, "rust-ida_amd", "csrc")


def build(tmp_path):
    exe = str(tmp_path / "c_consumer")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"), "-o", exe,
                           os.path.join(ROOT, "tests", "native", "c_consumer.c"), "-L", CSRC, "-lidahip", "-Wl,-rpath," + CSRC])
    return exe


def test_c_consumer_compiles_and_links_as_c99(tmp_path):
    exe = build(tmp_path)
    out = subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
    assert "libidahip.so" in out
    # the ensemble header is C as well
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"),
                           "-x", "c", os.path.join(ROOT, "include", "ida_ensemble.h")])


@pytest.mark.gpu
def test_c_consumer_reproduces_the_reference_goldens(tmp_path):
    G = json.load(open(os.path.join(GD, "dense_goldens.json")))
    W = json.load(open(os.path.join(GD, "wrms_golden.json")))
    b = lambda k: dict(G[k]["bindings"])
    colmajor = lambda m: np.asarray(m).T.ravel()  # the goldens are printed row by row; the ABI is column-major (dense.rs:222)
    mats = [b("test_get_rf1")["mat_a"], b("test_get_rf2")["mat_a"]]
    lus = [b("test_get_rs1")["mat_a"], b("test_get_rs2")["mat_a"]]
    piv = [b("test_get_rs1")["pivot"], b("test_get_rs2")["pivot"]]
    rhs = [b("test_get_rs1")["b"], b("test_get_rs2")["b"]]
    x = [[W["x"]] * 3, [1.0, -2.0, 3.0]]
    w = [[W["w"]] * 3, [0.5, 0.25, 2.0]]
    lines = [np.concatenate([colmajor(m) for m in mats]), np.concatenate([colmajor(m) for m in lus]), np.ravel(piv), np.ravel(rhs),
             np.ravel(x), np.ravel(w)]
    stdin = "\n".join(" ".join(float(v).hex() for v in line) for line in lines) + "\n"
    out = subprocess.run([build(tmp_path)], input=stdin, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    res = {l.split()[0]: l.split()[1:] for l in out.stdout.splitlines()}
    lu = np.array([float.fromhex(v) for v in res["lu"]]).reshape(2, 3, 3).transpose(0, 2, 1)
    assert np.array_equal(lu, np.array([b("test_get_rf1")["expect"], b("test_get_rf2")["expect"]]))   # assert_eq! (dense.rs:287,310)
    assert [int(v) for v in res["piv"]] == [2, 1, 2, 2, 1, 2] and res["info"] == ["0", "0"]
    sol = np.array([float.fromhex(v) for v in res["x"]]).reshape(2, 3)
    assert np.array_equal(sol, np.array([b("test_get_rs1")["expect"], b("test_get_rs2")["expect"]]))  # dense.rs:238,264
    nrm = [float.fromhex(v) for v in res["wrms"]]
    assert nrm[0] == W["expect"] == 0.25                                                              # norm_rms.rs:64-70
    p = [1.0 * 0.5, -2.0 * 0.25, 3.0 * 2.0]
    assert nrm[1] == float(np.sqrt(((p[0] * p[0] + p[1] * p[1]) + p[2] * p[2]) / 3.0))
