"""glibc_pow::pow (rust-ida_amd/csrc/glibc_pow.hpp), the pow of the device-resident step-size controller, against glibc.
The reference's controller calls f64::powf = the platform libm's pow (/root/reference/src/lib.rs:1163-1169,
src/impl_complete_step.rs:128-132, src/ida_nls.rs:249-253; exactness pinned by src/tests/complete_step.rs through
tests/test_oracle_stepper.py); the device may only take over that controller if its pow returns the same bits.
CPU: the header compiled for the host against libm on 10^8 random arguments, special values and the committed fixture.
GPU: the device build against the fixture (2^20 results of the build container's libm) and against the GPU box's own libm."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

from pow_cases import pow_cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "glibc_pow_2p20.npz")


def libm_pow(x, y):
    libm = ctypes.CDLL("libm.so.6")
    libm.pow.restype = ctypes.c_double
    libm.pow.argtypes = [ctypes.c_double, ctypes.c_double]
    return np.array([libm.pow(float(a), float(b)) for a, b in zip(x, y)])


def test_fixture_is_this_machines_libm():
    """The committed results are what this machine's libm computes (a CPU without FMA would select another variant of glibc's
    pow; the device code restates the FMA variant)."""
    x, y = pow_cases(1 << 20)
    want = np.load(GOLD)["bits"]
    sel = np.arange(0, 1 << 20, 97)
    assert np.array_equal(libm_pow(x[sel], y[sel]).view(np.uint64), want[sel])


def test_host_build_of_the_header_equals_libm(tmp_path):
    exe = str(tmp_path / "pow_check")
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-I", os.path.join(ROOT, "rust-ida_amd", "csrc"), "-o", exe,
                           os.path.join(ROOT, "tests", "native", "pow_check.cpp")])
    out = subprocess.run([exe, "100000000"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-2000:]
    assert "100000000 cases, 0 mismatches" in out.stdout


def test_tables_are_the_ones_in_libm(tmp_path):
    """tools/extract_glibc_pow_tables.py run again gives the committed header (the tables come from the libm next to us)."""
    libm = "/lib/x86_64-linux-gnu/libm.so.6"
    if not os.path.exists(libm):
        pytest.skip("no libm at the expected path")
    committed = open(os.path.join(ROOT, "rust-ida_amd", "csrc", "glibc_pow_tables.hpp")).read()
    env = dict(os.environ)
    script = open(os.path.join(ROOT, "tools", "extract_glibc_pow_tables.py")).read().replace(
        'os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rust-ida_amd", "csrc", "glibc_pow_tables.hpp")',
        repr(str(tmp_path / "t.hpp")))
    p = tmp_path / "extract.py"
    p.write_text(script)
    subprocess.check_call([sys.executable, str(p), libm], env=env)
    strip = lambda s: "\n".join(l for l in s.splitlines() if not l.startswith("// GENERATED"))
    assert strip(open(tmp_path / "t.hpp").read()) == strip(committed)


@pytest.mark.gpu
def test_device_pow_equals_glibc():
    import idahip
    ctx = idahip.Ctx("lorenz63", 3, 1)
    x, y = pow_cases(1 << 20)
    got = ctx.pow_batch(x, y).view(np.uint64)
    want = np.load(GOLD)["bits"]
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, (bad[:5], x[bad[:5]], y[bad[:5]])
    # this box's own libm on further arguments, general exponents included
    x2, y2 = pow_cases(1 << 16, seed=7)
    y2 = y2 * np.linspace(0.1, 9.0, y2.size)
    assert np.array_equal(ctx.pow_batch(x2, y2).view(np.uint64), libm_pow(x2, y2).view(np.uint64))
    sp = np.array([0.0, 1.0, 2.0, 0.5, np.inf, np.nan, 5e-324, 1e-310, 1e300, 1e-300, 1.0000000000000002, 0.9999999999999999])
    xs, ys = np.meshgrid(sp, np.array([0.5, -0.5, 1.0, 1.0 / 3.0, -0.2, 2.0, 0.0, np.inf, np.nan]))
    g, w = ctx.pow_batch(xs.ravel(), ys.ravel()), libm_pow(xs.ravel(), ys.ravel())
    assert np.array_equal(np.isnan(g), np.isnan(w)) and np.array_equal(g[~np.isnan(g)].view(np.uint64), w[~np.isnan(w)].view(np.uint64))
    ctx.close()
