"""the LSolverType-dependent branch of idaLsSolve (/root/reference/src/ida_ls.rs:316-418, LSolverType at
/root/reference/crates/linear/src/lib.rs:15-20) as restated in rust-ida_amd/host/ida_controller.hpp (idactl::lsolve_tol,
after_lsolve), which the host stepper and both device steppers call around every linear solve.
CPU: the two functions for all three solver types against the reference's text. GPU: the library's solver is the dense
direct one (dense.rs:30-36: get_type = Direct, num_iters = 0, res_norm = 0), and after a config-3 integration the
iterative-solver counters nli / ncfl are still zero on every stepper."""
import math
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_lsolve_tol_and_after_lsolve_follow_the_reference_text(tmp_path):
    exe = str(tmp_path / "lsolve_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-I", os.path.join(ROOT, "rust-ida_amd", "host"), "-o", exe,
                           os.path.join(ROOT, "tests", "native", "lsolve_check.cpp")])
    out = subprocess.run([exe], capture_output=True, text=True, check=True).stdout.strip().splitlines()
    assert len(out) == 12
    DIRECT, ITERATIVE, MATRIX_ITERATIVE = 0, 1, 2  # include/ida_hip.h idahip_ls_kind = LSolverType's order
    for line in out:
        m = re.match(r"type (\d) cjratio (\S+) failed (\d) tol (\S+) nli (\d+) ncfl (\d+) scale (\d)", line)
        t, cjratio, failed, tol, nli, ncfl, scale = int(m[1]), float(m[2]), int(m[3]), float(m[4]), int(m[5]), int(m[6]), int(m[7])
        # ida_ls.rs:323-329: tol = sqrt_n * eplifac for Iterative | MatrixIterative, zero otherwise; eplifac = 0.05 (:211)
        assert tol == (0.0 if t == DIRECT else math.sqrt(512.0) * 0.05)
        # :389-400: nli += num_iters for the iterative types only (started at 10, the solver reported 7)
        assert nli == (10 if t == DIRECT else 17)
        # :413-415: ncfl += 1 when the solver returned an error, whatever its type (started at 3)
        assert ncfl == 3 + failed
        # :405-410: the correction is scaled by 2 / (1 + cjratio) for Direct | MatrixIterative, and only if cjratio != 1
        assert scale == (1 if (t in (DIRECT, MATRIX_ITERATIVE) and cjratio != 1.0) else 0)


@pytest.mark.gpu
@pytest.mark.parametrize("device_ctl", [1, 0])
def test_the_librarys_solver_is_direct_and_the_iterative_counters_stay_zero(device_ctl):
    import idahip
    from idahip import problems
    p = problems.linear_dense(n=96, batch=24, procs=1)
    ctx = problems.make_ctx(p)
    assert ctx.H.idahip_ls_type(ctx.h) == 0       # IDAHIP_LS_DIRECT (dense.rs:30-32)
    assert ctx.H.idahip_ls_num_iters(ctx.h) == 0  # traits.rs:82-86 default
    assert ctx.H.idahip_ls_res_norm(ctx.h) == 0.0
    ens = idahip.Ensemble(ctx, p["yy0"], p["yp0"])
    ens.set_device_controller(device_ctl)
    assert ens.device_controller_active() == (2 if device_ctl else 0)
    status, _, reached = ens.solve_schedule(p["touts"][:4])
    assert (status == 0).all() and (reached == 4).all()
    c = ens.counters()
    assert c["nni"].sum() > 0 and c["nsetups"].sum() > 0
    assert not c["nli"].any() and not c["ncfl"].any()
    # the 2 / (1 + cjratio) scaling of the Direct branch was exercised: some Newton solve ran on a stale Jacobian
    assert (c["nni"] > c["nsetups"]).any()
    ens.close()
    ctx.close()
