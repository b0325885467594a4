"""GPU parity of the whole hot path: the batched HIP Newton/LU/norm kernels driven by the host stepper
(libidaens) against the CPU oracle integrating the same IVPs one by one.

Bar (BASELINE.json north_star): step-accept / order counts bit-exact; outputs within fp64 tolerance. Because every
device sum that feeds a decision is accumulated in the reference's order and no FMA is contracted, the outputs are
in fact bit-identical, and the tests assert that (np.array_equal) -- a stronger statement than the tolerance
north_star allows (rtol 1e-12 would already pass its bar).
"""
import json
import os

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
CNT = ("nst", "nre", "nje", "nsetups", "nni", "netf", "ncfn", "n_attempts")


def run_gpu(prob, touts=None, lu_variant=4):
    import idahip
    from idahip import problems
    ctx = problems.make_ctx(prob)
    ctx.set_lu_variant(lu_variant)
    ens = idahip.Ensemble(ctx, prob["yy0"], prob["yp0"])
    touts = prob["touts"] if touts is None else touts
    yy, yp = [], []
    for t in touts:
        status, tret = ens.solve(t)
        assert (status == 0).all(), status
        assert np.array_equal(tret, np.full_like(tret, t))
        yy.append(ens.yy())
        yp.append(ens.yp())
    return ens, np.array(yy), np.array(yp)


def run_oracle(prob, touts=None, nthreads=8):
    touts = prob["touts"] if touts is None else touts
    return O.run_ensemble(prob["kind"], prob["n"], prob["yy0"], prob["yp0"], prob["rtol"], prob["atol"], touts,
                          params=prob.get("params"), A=prob.get("A"), B=prob.get("B"), c=prob.get("c"), nthreads=nthreads)


def check(prob, touts=None, lu_variant=4):
    ens, yy, yp = run_gpu(prob, touts, lu_variant)
    ref = run_oracle(prob, touts)
    assert (ref["status"] == 0).all()
    c = ens.counters()
    for k in CNT:
        assert np.array_equal(c[k], ref["counters"][k]), (k, c[k], ref["counters"][k])
    assert np.array_equal(c["kused"], ref["kused"])
    assert np.array_equal(ens.real("hused"), ref["hused"])
    assert np.array_equal(yy, ref["yy"])
    assert np.array_equal(yp, ref["yp"])
    return ens, ref


def test_roberts_batch_matches_reference_run():
    """Config 1 through the GPU path (no root finding there): same 362 steps / 377 attempts / 537 Newton iterations
    and the exact bits of y(4e10) of the reference-validated oracle run."""
    from idahip import problems
    p = problems.roberts()
    p["yy0"] = np.tile(p["yy0"], (5, 1))
    p["yp0"] = np.tile(p["yp0"], (5, 1))
    ens, yy, yp = run_gpu(p)
    c = ens.counters()
    assert (c["nst"] == 362).all() and (c["n_attempts"] == 377).all() and (c["nni"] == 537).all()
    assert (c["nre"] == 537).all() and (c["nje"] == 60).all() and (c["netf"] == 15).all() and (c["ncfn"] == 0).all()
    assert (c["nls_nconvfails"] == 5).all()
    assert [v.hex() for v in yy[-1, 0]] == ["0x1.a1d277a766cb0p-25", "0x1.b61e4814ea4bbp-43", "0x1.fffffe5e2d1adp-1"]
    R = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "roberts_example.json")))
    ref = np.array(R["reference_solution_t4e10"])
    ewt = 1.0 / (R["rtol"] * np.abs(ref) + 10.0 * np.array(R["atol"]))
    assert O.wrms(yy[-1, 3] - ref, ewt) < 1.0  # check_ans, examples/roberts.rs:9-51


def test_lorenz63_ensemble():
    from idahip import problems
    check(problems.lorenz63(batch=96), touts=0.1 * np.arange(1, 21))


@pytest.mark.parametrize("variant", [3, 4])
@pytest.mark.parametrize("n,batch", [(12, 6), (33, 5), (64, 8), (100, 4), (192, 3)])
def test_linear_dense_ensemble(n, batch, variant):
    from idahip import problems
    p = problems.linear_dense(n=n, batch=batch)
    ens, ref = check(p, lu_variant=variant)
    assert (ref["counters"]["nsetups"] > 1).all()  # stale-Jacobian iterations and re-factorisations both occurred


def test_linear_dense_trace_of_one_system():
    import idahip
    from idahip import problems
    p = problems.linear_dense(n=48, batch=3)
    ctx = problems.make_ctx(p)
    ens = idahip.Ensemble(ctx, p["yy0"], p["yp0"])
    ens.trace_system(1)
    for t in p["touts"]:
        ens.solve(t)
    tr = ens.trace()
    o = O.OracleIda("linear_dense", 48, p["yy0"][1], p["yp0"][1], p["rtol"], p["atol"], A=p["A"][1], B=p["B"][1], c=p["c"][1])
    O.lib().oracle_ida_record_steps(o.h, 1)
    for t in p["touts"]:
        assert o.solve(t)[0] == 0
    rec = o.recorded_steps()
    assert np.array_equal(tr, rec[:, :3])  # every accepted step: same t_n, h_used, order -- bit for bit


@pytest.mark.parametrize("variant", [3, 4])
@pytest.mark.parametrize("n,batch", [(40, 4), (130, 3)])
def test_heat1d_ensemble(n, batch, variant):
    from idahip import problems
    check(problems.heat1d(n=n, batch=batch), lu_variant=variant)


def test_one_step_mode_and_interpolation_past_tout():
    import idahip
    from idahip import problems
    p = problems.linear_dense(n=16, batch=2)
    ctx = problems.make_ctx(p)
    ens = idahip.Ensemble(ctx, p["yy0"], p["yp0"])
    st, tret = ens.solve(1.0, itask=1)  # OneStep
    assert (st == 0).all() and (ens.counter("nst") == 1).all() and np.array_equal(tret, ens.real("tn"))
    o = [O.OracleIda("linear_dense", 16, p["yy0"][s], p["yp0"][s], p["rtol"], p["atol"], A=p["A"][s], B=p["B"][s], c=p["c"][s])
         for s in range(2)]
    for s in range(2):
        so, to = o[s].solve(1.0, itask=1)
        assert so == 0 and to == tret[s]
        assert np.array_equal(o[s].getv("yy"), ens.yy()[s])
    # Normal mode to t=0.5, then a tout already passed -> pure interpolation (stop_test1)
    ens.solve(0.5)
    t2 = float(np.min(ens.real("tn") - 0.25 * ens.real("hused")))  # inside every system's last step
    st, tret = ens.solve(t2)
    yy, yp = ens.yy(), ens.yp()
    for s in range(2):
        o[s].solve(0.5)
        so, to = o[s].solve(t2)
        assert so == st[s] == 0 and to == tret[s] == t2
        assert np.array_equal(o[s].getv("yy"), yy[s]) and np.array_equal(o[s].getv("yp"), yp[s])
    # a tout before the last step is rejected by both (IdaError::BadTimeValue)
    st, _ = ens.solve(0.01)
    assert (st == -26).all() and all(o[s].solve(0.01)[0] == -26 for s in range(2))


def test_round_limited_solve_resumes_identically():
    import idahip
    from idahip import problems
    p = problems.linear_dense(n=24, batch=4)
    a = idahip.Ensemble(problems.make_ctx(p), p["yy0"], p["yp0"])
    b = idahip.Ensemble(problems.make_ctx(p), p["yy0"], p["yp0"])
    a.solve(0.3)
    while True:
        st, _ = b.solve(0.3, max_rounds=3)
        if (st != 99).all():
            break
    assert np.array_equal(a.yy(), b.yy()) and np.array_equal(a.counter("nni"), b.counter("nni"))


def test_mxstep_is_reported_per_system():
    import idahip
    from idahip import problems
    p = problems.linear_dense(n=16, batch=2)
    ens = idahip.Ensemble(problems.make_ctx(p), p["yy0"], p["yp0"])
    ens.set_max_num_steps(5)
    st, tret = ens.solve(1.0)
    assert (st == -1).all() and (ens.counter("nst") == 5).all()  # IDA_TOO_MUCH_WORK after mxstep steps


@pytest.mark.parametrize("kind,n,batch", [("linear_dense", 64, 5), ("linear_dense", 33, 3), ("heat1d", 40, 3), ("lorenz63", 3, 7)])
def test_sys_setup_in_one_call_equals_sys_then_lsetup(kind, n, batch):
    """idahip_nls_sys_setup (sys + setup in one device call; for the linear dense problem J = B + cj A falls out of the
    residual pass) leaves the same residual, factors, pivots and state as idahip_nls_sys followed by idahip_nls_lsetup."""
    import idahip
    from idahip import problems
    p = {"linear_dense": lambda: problems.linear_dense(n=n, batch=batch), "heat1d": lambda: problems.heat1d(n=n, batch=batch),
         "lorenz63": lambda: problems.lorenz63(batch=batch)}[kind]()
    rng = np.random.default_rng(n)
    tn, cj = 0.125, 40.0 + rng.uniform(size=batch)
    idx = np.arange(batch - 1, -1, -1) if batch > 3 else np.arange(batch)   # a permuted list
    ypred = p["yy0"] + 1e-3 * rng.standard_normal(p["yy0"].shape)
    ee = 1e-2 * rng.standard_normal(p["yy0"].shape)
    for reset_ee in (True, False):
        out = []
        for fused in (False, True):
            ctx = problems.make_ctx(p)
            ctx.upload(idahip.F_YYPREDICT, ypred)
            ctx.upload(idahip.F_YPPREDICT, p["yp0"])
            ctx.upload(idahip.F_EE, ee)
            if fused:
                rc, info = ctx.nls_sys_setup(tn, cj[idx], reset_ee=reset_ee, idx=idx)
            else:
                ctx.nls_sys(tn, cj[idx], reset_ee=reset_ee, idx=idx)
                rc, info = ctx.nls_lsetup(tn, cj[idx], idx=idx)
            assert rc == 0 and not info.any()
            lus = [ctx.download_lu(s) for s in range(batch)]
            out.append((ctx.download(idahip.F_DELTA), ctx.download(idahip.F_SAVRES), ctx.download(idahip.F_YY),
                        ctx.download(idahip.F_YP), ctx.download(idahip.F_EE), lus))
        for a, b in zip(out[0][:5], out[1][:5]):
            assert np.array_equal(a, b)
        for (lu_a, piv_a), (lu_b, piv_b) in zip(out[0][5], out[1][5]):
            assert np.array_equal(lu_a, lu_b) and np.array_equal(piv_a, piv_b)


def test_singular_jacobian_fails_like_the_oracle_and_spares_the_batch():
    """One system whose Newton matrix B + cj A has an exactly zero column for every cj: its linear setup fails at every
    attempt (recoverable, the step size is cut, then the step fails for good); the oracle walks the same path, and the
    other systems of the batch are untouched by it."""
    import idahip
    from idahip import problems
    n, batch, bad = 12, 4, 2
    p = problems.linear_dense(n=n, batch=batch)
    p["A"][bad, 5, :] = 0.0   # storage is [col][row]: column 5 of A and of B
    p["B"][bad, 5, :] = 0.0
    ctx = problems.make_ctx(p)
    ens = idahip.Ensemble(ctx, p["yy0"], p["yp0"])
    status, tret = ens.solve(0.1)
    ref = run_oracle(p, touts=[0.1])
    assert np.array_equal(status, ref["status"])
    assert status[bad] < 0 and (np.delete(status, bad) == 0).all()
    c = ens.counters()
    for k in CNT:
        assert np.array_equal(c[k], ref["counters"][k]), k
    good = np.arange(batch) != bad
    assert np.array_equal(ens.yy()[good], ref["yy"][0][good])
    # the healthy systems equal a run without the bad one
    q = {k: (v[good] if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[0] == batch else v) for k, v in p.items()}
    ens2 = idahip.Ensemble(problems.make_ctx(q), q["yy0"], q["yp0"])
    ens2.solve(0.1)
    assert np.array_equal(ens2.yy(), ens.yy()[good])


def test_newton_convergence_failures_follow_c_ida_like_the_oracle():
    """Lorenz at r = 400 with very loose tolerances: some steps are so long that Newton does not converge in four
    iterations even with a fresh Jacobian (ncfn > 0). This is quirk Q3/Q4 of SURVEY.md 9: the reference's own text would loop
    (newton.rs:146-153) or treat the failure as fatal (src/lib.rs:1133-1140); oracle and product follow C IDA -- the step is
    repeated with h/4 -- and the test pins product == oracle on that path (state and every counter), not product ==
    reference. The counter `nconv_jcur` says how often a system took it; the BASELINE configurations never do
    (tests/test_gpu_fullsize.py, bench.py's reference_text_paths)."""
    from idahip import problems
    p = problems.lorenz63(batch=512)
    pr, rr, bb = 10.0, 400.0, 8.0 / 3.0
    p["params"][:, 1] = rr
    y0 = p["yy0"]
    p["yp0"] = np.stack([pr * (y0[:, 1] - y0[:, 0]), y0[:, 0] * (rr - y0[:, 2]) - y0[:, 1], y0[:, 0] * y0[:, 1] - bb * y0[:, 2]], axis=1)
    p["rtol"], p["atol"] = 0.3, np.array([0.1])
    ens, ref = check(p, touts=p["touts"][:20])
    assert ref["counters"]["ncfn"].sum() > 0 and ref["counters"]["netf"].sum() > 0
    assert ens.counter("nconv_jcur").sum() > 0 and np.array_equal(ens.counter("nconv_jcur") + ens.counter("nlufail"), ens.counter("ncfn"))


@pytest.mark.parametrize("maxord", [1, 2, 3])
def test_max_order_limit(maxord):
    """Ida::set_max_ord: the order selection is capped (complete_step / test_error branches on kk == maxord), steps and
    state still bit-identical to the oracle with the same cap (a low cap needs many more steps: mxstep is raised on both
    sides)."""
    import idahip
    from idahip import problems
    p = problems.linear_dense(n=20, batch=3)
    touts = p["touts"][:2]
    ens = idahip.Ensemble(problems.make_ctx(p), p["yy0"], p["yp0"])
    ens.set_max_ord(maxord)
    ens.set_max_num_steps(200000)
    for t in touts:
        st, _ = ens.solve(float(t))
        assert (st == 0).all()
    assert (ens.counter("kused") <= maxord).all()
    for s in range(3):
        o = O.OracleIda("linear_dense", 20, p["yy0"][s], p["yp0"][s], p["rtol"], p["atol"], A=p["A"][s], B=p["B"][s], c=p["c"][s])
        o.set("maxord", maxord)
        o.set("mxstep", 200000)
        for t in touts:
            assert o.solve(float(t))[0] == 0
        assert np.array_equal(o.getv("yy"), ens.yy()[s]) and np.array_equal(o.getv("yp"), ens.yp()[s])
        assert o.get("nst") == ens.counter("nst")[s] and o.get("nni") == ens.counter("nni")[s]


def test_roberts_example_with_root_finding():
    """examples/roberts.rs as the reference runs it: 12 outputs at 0.4 * 10^k with the two root functions (y1 - 1e-4,
    y3 - 0.01) active. Every return of Ida::solve -- status (0 or IDA_ROOT_RETURN), t_ret, y, rootsfound -- and the
    final counters (362 steps, 404 root-function evaluations) equal the oracle's, which is pinned on the reference's
    own figures (tests/test_oracle_roberts.py); five copies run as one batch."""
    import idahip
    from idahip import problems
    R = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "roberts_example.json")))
    p = problems.roberts()
    B = 5
    p["yy0"] = np.tile(p["yy0"], (B, 1))
    p["yp0"] = np.tile(p["yp0"], (B, 1))
    ens = idahip.Ensemble(problems.make_ctx(p), p["yy0"], p["yp0"])
    ens.set_roots([0, 2], [0.0001, 0.01])
    ida = O.OracleIda("roberts", 3, R["yy0"], R["yp0"], R["rtol"], R["atol"])
    tout, iout, nroot_returns = R["tout0"], 0, 0
    while iout < R["nout"]:
        st_o, tret_o = ida.solve(tout)
        st, tret = ens.solve(tout)
        assert (st == st_o).all() and st_o >= 0
        assert np.array_equal(tret, np.full(B, tret_o))
        assert np.array_equal(ens.yy(), np.tile(ida.getv("yy"), (B, 1)))
        assert np.array_equal(ens.yp(), np.tile(ida.getv("yp"), (B, 1)))
        if st_o == 2:
            nroot_returns += 1
            assert np.array_equal(ens.roots_found(), np.tile(ida.getv("iroots").astype(np.int32), (B, 1)))
        else:
            iout += 1
            tout *= R["tout_factor"]
    assert nroot_returns == 2
    c = ens.counters()
    assert (c["nst"] == 362).all() and (c["n_attempts"] == 377).all() and (c["nge"] == 404).all()
    assert (c["nni"] == 537).all() and (c["netf"] == 15).all()
    # config 1 and the paths where oracle and product follow C IDA instead of the reference's text: never taken
    # (the five Newton-internal re-setups are the reference's own path, newton.rs:146-152)
    for k in ("ncfn", "nlufail", "nconv_jcur", "nfail_first"):
        assert (c[k] == 0).all(), k
    assert (c["nls_nconvfails"] == 5).all()


def test_user_root_function_through_the_host_callback():
    """idaens_set_root_fn: the Root trait for any user function (src/traits.rs:72-90), evaluated on the host with y(t), y'(t)
    interpolated on the device. (1) The Roberts example's two functions written as a callback reproduce the built-in family
    return for return (status, t_ret, y, rootsfound, 404 evaluations). (2) A function of t alone, g = t - 0.5 on Lorenz63,
    stops every system at t = 0.5 to within the bracketing tolerance. (3) A callback that raises fails only its own system
    with IDAENS_RTFUNC_FAIL and the exception is kept."""
    import idahip
    from idahip import problems
    R = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "roberts_example.json")))
    p = problems.roberts()
    B = 3
    p["yy0"] = np.tile(p["yy0"], (B, 1))
    p["yp0"] = np.tile(p["yp0"], (B, 1))
    fam = idahip.Ensemble(problems.make_ctx(p), p["yy0"], p["yp0"])
    fam.set_roots([0, 2], [0.0001, 0.01])
    cb = idahip.Ensemble(problems.make_ctx(p), p["yy0"], p["yp0"])
    calls = []

    def g(sys, t, yy, yp):
        calls.append(sys)
        return [yy[0] - 0.0001, yy[2] - 0.01]

    cb.set_root_fn(2, g)
    tout, iout, nroot_returns = R["tout0"], 0, 0
    while iout < R["nout"]:
        st_f, tret_f = fam.solve(tout)
        st, tret = cb.solve(tout)
        assert np.array_equal(st, st_f) and np.array_equal(tret, tret_f)
        assert np.array_equal(cb.yy(), fam.yy()) and np.array_equal(cb.yp(), fam.yp())
        if st_f[0] == 2:
            nroot_returns += 1
            assert np.array_equal(cb.roots_found(), fam.roots_found())
        else:
            iout += 1
            tout *= R["tout_factor"]
    assert nroot_returns == 2 and (cb.counter("nge") == 404).all() and len(calls) == 404 * B
    fam.close()
    cb.close()

    q = problems.lorenz63(batch=6)
    ens = idahip.Ensemble(problems.make_ctx(q), q["yy0"], q["yp0"])
    ens.set_root_fn(1, lambda sys, t, yy, yp: [t - 0.5])
    st, tret = ens.solve(1.0)
    assert (st == 2).all() and np.all(np.abs(tret - 0.5) < 1e-9)
    assert (ens.roots_found() == -1).all()  # the reference reports glo.signum() (impl_r_check.rs:404): -1 for an increasing g
    st, tret = ens.solve(1.0)  # past the root: the call now reaches tout
    assert (st == 0).all() and np.array_equal(tret, np.full(6, 1.0))
    ens.close()

    ens = idahip.Ensemble(problems.make_ctx(q), q["yy0"], q["yp0"])

    def bad(sys, t, yy, yp):
        if sys == 4:
            raise ValueError("no g for system 4")
        return [yy[0] - 1.0e9]

    ens.set_root_fn(1, bad)
    st, tret = ens.solve(0.2)
    assert st[4] == -12 and (np.delete(st, 4) == 0).all()
    assert isinstance(ens.ctx.take_callback_error(), ValueError)
    assert ens.ctx.take_callback_error() is None  # handed out once: not attached to a later, unrelated error
    ens.close()


@pytest.mark.parametrize("kind", ["linear_dense", "lorenz63", "heat1d"])
def test_output_schedule_equals_sequential_solve_calls(kind):
    """idaens_solve_schedule: every system runs solve(t1), solve(t2), ... without waiting for the others. The output at
    every tout, the final state and every counter equal the oracle's sequential calls; far fewer lock-step rounds."""
    import idahip
    from idahip import problems
    p = {"linear_dense": lambda: problems.linear_dense(n=40, batch=9), "lorenz63": lambda: problems.lorenz63(batch=300),
         "heat1d": lambda: problems.heat1d(n=40, batch=4)}[kind]()
    touts = p["touts"][:10]
    ref = run_oracle(p, touts=touts)
    ens = idahip.Ensemble(problems.make_ctx(p), p["yy0"], p["yp0"])
    status, tret, reached, yo, ypo = ens.solve_schedule(touts, outputs=True)
    assert (status == 0).all() and (reached == len(touts)).all() and np.array_equal(tret, np.full_like(tret, touts[-1]))
    assert np.array_equal(yo, ref["yy"]) and np.array_equal(ypo, ref["yp"])
    assert np.array_equal(ens.yy(), ref["yy"][-1]) and np.array_equal(ens.yp(), ref["yp"][-1])
    c = ens.counters()
    for k in CNT:
        assert np.array_equal(c[k], ref["counters"][k]), k
    # the same through solve() call by call costs more rounds (every call waits for its slowest system)
    seq = idahip.Ensemble(problems.make_ctx(p), p["yy0"], p["yp0"])
    for t in touts:
        seq.solve(float(t))
    assert np.array_equal(seq.yy(), ens.yy())
    assert ens.total_rounds() <= seq.total_rounds()
    # sliced by a round limit it continues where it stopped
    sl = idahip.Ensemble(problems.make_ctx(p), p["yy0"], p["yp0"])
    while True:
        st, _, rc = sl.solve_schedule(touts, max_rounds=2)
        if (st != 99).all():
            break
    assert (st == 0).all() and (rc == len(touts)).all()
    assert np.array_equal(sl.yy(), ens.yy()) and np.array_equal(sl.counter("nni"), ens.counter("nni"))
    assert sl.total_rounds() == ens.total_rounds()


def test_streaming_restarts_reproduce_fresh_integrations():
    """idaens_stream: systems that finish t = 1 are created anew and start over while the others keep going. Whatever
    pass a system is in, its accepted steps are those of a fresh integration: after any number of rounds its (nst, t_n,
    h_used, order) is exactly the oracle's nst-th step."""
    import idahip
    from idahip import problems
    n, B = 24, 12
    p = problems.linear_dense(n=n, batch=B)
    ens = idahip.Ensemble(problems.make_ctx(p), p["yy0"], p["yp0"])
    steps = []
    for s in range(B):
        o = O.OracleIda("linear_dense", n, p["yy0"][s], p["yp0"][s], p["rtol"], p["atol"], A=p["A"][s], B=p["B"][s], c=p["c"][s])
        O.lib().oracle_ida_record_steps(o.h, 1)
        for t in p["touts"]:
            assert o.solve(float(t))[0] == 0
        steps.append(o.recorded_steps())
    full_nni = run_oracle(p)["counters"]["nni"]
    rounds, passes = 0, 0
    while passes < 3 * B:
        passes = ens.stream(p["touts"], 7)
        rounds += 7
        nst, tn, hu, ku = ens.counter("nst"), ens.real("tn"), ens.real("hused"), ens.counter("kused")
        for s in range(B):
            if nst[s] > 0:
                assert np.array_equal(steps[s][nst[s] - 1, :3], [tn[s], hu[s], float(ku[s])]), (s, nst[s])
    assert ens.total_rounds() == rounds
    # every completed integration contributed its full Newton count, the running ones their partial counts
    assert ens.total_newton_iters() >= 3 * full_nni.min() * B // 2


@pytest.mark.parametrize("kind", ["linear_dense", "lorenz63"])
def test_get_dky_matches_the_oracle(kind):
    """Ida::get_dky (src/lib.rs:424-529) for every system and every k = 0 .. kused: coefficients on the host (libidaens), sums
    on the device (idahip_get_dky); bit-identical to the oracle's restatement, argument checks included."""
    import idahip
    from idahip import problems
    p = problems.linear_dense(n=48, batch=7) if kind == "linear_dense" else problems.lorenz63(batch=40)
    touts = [float(t) for t in p["touts"][:3]]
    ens = idahip.Ensemble(problems.make_ctx(p), p["yy0"], p["yp0"])
    for t in touts:
        status, _ = ens.solve(t)
        assert (status == 0).all()
    B, n = p["yy0"].shape
    tn, hused, kused = ens.real("tn"), ens.real("hused"), ens.counter("kused")
    oracles = []
    for s in range(B):
        kw = {"params": p["params"][s]} if kind == "lorenz63" else {"A": p["A"][s], "B": p["B"][s], "c": p["c"][s]}
        o = O.OracleIda(kind, n, p["yy0"][s], p["yp0"][s], p["rtol"], p["atol"], **kw)
        for t in touts:
            assert o.solve(t)[0] == 0
        assert o.get("tn") == tn[s] and int(o.get("kused")) == kused[s]
        oracles.append(o)
    # touts[-1] lies inside the last step of every system (all have stepped past it); a little earlier it still does for
    # most and is IDA_BAD_T for the others -- both outcomes must equal the oracle's
    for t, k in [(tt, kk) for tt in (touts[-1], touts[-1] - 0.37 * hused.min()) for kk in range(0, int(kused.max()) + 2)]:
        status, dky = ens.get_dky(t, k)
        for s in range(B):
            st_o, d_o = oracles[s].get_dky(t, k)
            assert status[s] == st_o, (k, s)
            if st_o == 0:
                assert np.array_equal(dky[s], d_o), (k, s)
            else:
                assert st_o in (-25, -26) and (st_o == -26 or k > kused[s]) and np.isnan(dky[s]).all()
    status, _ = ens.get_dky(float(tn.min() - 3.0 * hused.max()), 0)   # before the last step of every system
    assert (status == -26).all()
    ens.close()


def test_user_problem_through_host_callbacks_reproduces_the_roberts_example():
    """IDAHIP_HOST_CALLBACK: an arbitrary `IdaProblem` (src/traits.rs:12-70,92-94) whose res / jac are host functions. Roberts
    written as such callbacks (operation order of src/sample_problems/roberts.rs:47-91) runs the reference's example --
    12 outputs, two root functions -- with every return equal to the built-in device kernels' (i.e. the oracle's), and ends
    with the reference's figures: 362 steps, 377 attempts, 537 Newton iterations, the y(4e10) bits of SURVEY.md appendix A."""
    import idahip
    from idahip import problems
    R = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "roberts_example.json")))

    def res(sys, t, y, yp):
        r0 = -0.04 * y[0] + 1.0e4 * y[1] * y[2]
        r1 = -r0 - 3.0e7 * y[1] * y[1] - yp[1]
        r0 -= yp[0]
        return [r0, r1, y[0] + y[1] + y[2] - 1.0]

    def jac(sys, t, cj, y, yp, r):
        return [[-0.04 - cj, 1.0e4 * y[2], 1.0e4 * y[1]],
                [0.04, -1.0e4 * y[2] - 6.0e7 * y[1] - cj, -1.0e4 * y[1]],
                [1.0, 1.0, 1.0]]

    B = 2
    p = {"kind": "host_callback", "n": 3, "yy0": np.tile(R["yy0"], (B, 1)), "yp0": np.tile(R["yp0"], (B, 1)), "rtol": R["rtol"],
         "atol": np.array(R["atol"]), "res": res, "jac": jac}
    ens = idahip.Ensemble(problems.make_ctx(p), p["yy0"], p["yp0"])
    ens.set_roots([0, 2], [0.0001, 0.01])
    ida = O.OracleIda("roberts", 3, R["yy0"], R["yp0"], R["rtol"], R["atol"])
    tout, iout, nroot_returns = R["tout0"], 0, 0
    while iout < R["nout"]:
        st_o, tret_o = ida.solve(tout)
        st, tret = ens.solve(tout)
        assert (st == st_o).all() and st_o >= 0
        assert np.array_equal(tret, np.full(B, tret_o))
        assert np.array_equal(ens.yy(), np.tile(ida.getv("yy"), (B, 1)))
        if st_o == 2:
            nroot_returns += 1
        else:
            iout += 1
            tout *= R["tout_factor"]
    assert nroot_returns == 2
    c = ens.counters()
    assert (c["nst"] == 362).all() and (c["n_attempts"] == 377).all() and (c["nni"] == 537).all() and (c["netf"] == 15).all()
    y_end = ens.yy()[0]
    assert [float.hex(float(v)) for v in y_end] == ["0x1.a1d277a766cb0p-25", "0x1.b61e4814ea4bbp-43", "0x1.fffffe5e2d1adp-1"]
    ens.close()


def test_failing_user_callback_aborts_the_call_with_an_error():
    """A residual / Jacobian callback that fails (non-zero return; here: a Python exception caught by the binding's thunk)
    aborts the entry point that called it with -7 and a message naming the system; nothing unwinds across the C boundary. A
    newton_iter2 call on a host-callback ctx is refused (the fused iterations need a device residual)."""
    import idahip
    n, B = 3, 2
    calls = {"res": 0}

    def res(sys, t, y, yp):
        calls["res"] += 1
        if sys == 1:
            raise RuntimeError("user code failed")
        return [y[0] - 1.0, y[1] - 2.0, yp[2]]

    def jac(sys, t, cj, y, yp, r):
        raise RuntimeError("user code failed")

    ctx = idahip.Ctx("host_callback", n, B)
    with pytest.raises(idahip.IdaHipError, match="set_host_problem has not been called"):
        ctx.nls_sys(0.0, 1.0, True)
    ctx.set_host_problem(res, jac)
    ctx.set_tolerances(1e-6, 1e-8)
    ctx.nls_sys(0.0, 1.0, True, idx=[0])                        # system 0 alone is fine
    with pytest.raises(idahip.IdaHipError, match="residual function failed for system 1"):
        ctx.nls_sys(0.0, 1.0, True)
    with pytest.raises(idahip.IdaHipError, match="Jacobian function failed for system 0"):
        ctx.nls_lsetup(0.0, 1.0, idx=[0])
    assert calls["res"] >= 3


@pytest.mark.parametrize("kind", ["linear_dense", "lorenz63", "heat1d", "roberts"])
def test_fused_first_two_newton_iterations_change_nothing(kind):
    """SURVEY 8(f)-2, first slice: with idahip_newton_iter2 the first two Newton iterations of a solve and their convergence
    tests (ida_nls.rs:243-262, no powf needed for m <= 1) run in one device call. Every parity test of this file runs with
    it (the default); here the same integration with the switch off -- one host round trip per iteration, ctest on the
    host -- must give identical bits and counters, and the oracle's."""
    import idahip
    from idahip import problems
    p = {"linear_dense": lambda: problems.linear_dense(n=40, batch=9), "lorenz63": lambda: problems.lorenz63(batch=300),
         "heat1d": lambda: problems.heat1d(n=40, batch=4), "roberts": problems.roberts}[kind]()
    touts = p["touts"][:6]
    out = {}
    for fused in (1, 0):
        ens = idahip.Ensemble(problems.make_ctx(p), p["yy0"], p["yp0"])
        ens.set_fused_newton(fused)
        status, tret, reached = ens.solve_schedule(touts)
        assert (status == 0).all()
        out[fused] = (ens.yy(), ens.yp(), ens.counters(), ens.real("hused"))
        ens.close()
    assert np.array_equal(out[1][0], out[0][0]) and np.array_equal(out[1][1], out[0][1]) and np.array_equal(out[1][3], out[0][3])
    for k in out[1][2]:
        assert np.array_equal(out[1][2][k], out[0][2][k]), k
    ref = run_oracle(p, touts=touts)
    assert np.array_equal(out[1][0], ref["yy"][-1])
    for k in CNT:
        assert np.array_equal(out[1][2][k], ref["counters"][k]), k
