"""The device-resident stepper for small systems (idahip_tiny_solve: the whole of Ida::solve in one launch, step-size and
order controller on the device) against the lock-step host stepper (same controller source, platform pow)
and against the CPU oracle. Bar: bit-identical states, step sizes, orders and counters."""
import numpy as np
import pytest

import oracle_lib as O
from test_gpu_ensemble import CNT, run_oracle

pytestmark = pytest.mark.gpu


def make(prob, device_ctl):
    import idahip
    from idahip import problems
    ctx = problems.make_ctx(prob)
    ens = idahip.Ensemble(ctx, prob["yy0"], prob["yp0"])
    ens.set_device_controller(device_ctl)  # (raises if the library refuses: device pow != host pow)
    # the stepper that will really run: a 'device == host stepper' comparison must not silently compare host with host
    assert (ens.device_controller_active() != 0) == bool(device_ctl), ens.device_controller_active()
    return ctx, ens


def state(ens):
    c = ens.counters()
    return {**{k: c[k] for k in CNT + ("kused", "kk", "nls_nconvfails")}, "hused": ens.real("hused"), "hh": ens.real("hh"), "tn": ens.real("tn"),
            "yy": ens.yy(), "yp": ens.yp()}


def same(a, b):
    for k in a:
        assert np.array_equal(a[k], b[k]), k


def roberts_batch(batch=8):
    from idahip import problems
    p = problems.roberts()
    rng = np.random.Generator(np.random.PCG64(5))
    y0 = np.tile(p["yy0"], (batch, 1))
    y0[1:, 0] -= 1e-3 * rng.uniform(0, 1, batch - 1)   # consistent perturbation: y1 + y2 + y3 = 1 kept
    y0[1:, 2] = 1.0 - y0[1:, 0] - y0[1:, 1]
    yp0 = np.stack([-0.04 * y0[:, 0] + 1e4 * y0[:, 1] * y0[:, 2], 0.04 * y0[:, 0] - 1e4 * y0[:, 1] * y0[:, 2] - 3e7 * y0[:, 1] ** 2,
                    np.zeros(batch)], axis=1)
    yp0[:, 2] = -(yp0[:, 0] + yp0[:, 1])
    p.update(yy0=y0, yp0=yp0)
    return p


@pytest.mark.parametrize("name", ["lorenz63", "roberts"])
def test_device_stepper_equals_host_stepper_and_oracle(name):
    """Call by call (Ida::solve(tout) for every output time): device-resident stepper == host stepper == oracle."""
    from idahip import problems
    prob = problems.lorenz63(batch=192) if name == "lorenz63" else roberts_batch()
    touts = prob["touts"][:20] if name == "lorenz63" else prob["touts"]
    cd, dev = make(prob, 1)
    ch, host = make(prob, 0)
    ref = run_oracle(prob, touts)
    for i, t in enumerate(touts):
        sd, td = dev.solve(t)
        sh, th = host.solve(t)
        assert (sd == 0).all() and np.array_equal(sd, sh) and np.array_equal(td, th)
        same(state(dev), state(host))
        assert np.array_equal(dev.yy(), ref["yy"][i]) and np.array_equal(dev.yp(), ref["yp"][i])
    c = dev.counters()
    for k in CNT:
        assert np.array_equal(c[k], ref["counters"][k]), k
    assert np.array_equal(c["kused"], ref["kused"]) and np.array_equal(dev.real("hused"), ref["hused"])
    assert dev.total_rounds() == host.total_rounds()
    if name == "roberts":  # the reference's own run is system 0
        assert (c["nst"][0], c["n_attempts"][0], c["nni"][0], c["nsetups"][0], c["netf"][0]) == (362, 377, 537, 60, 15)


def test_schedule_outputs_and_round_limited_resume():
    """idaens_solve_schedule on the device: outputs at every tout, and a call cut into slices of 37 rounds gives the same."""
    from idahip import problems
    prob = problems.lorenz63(batch=96)
    touts = prob["touts"]
    cd, dev = make(prob, 1)
    ch, host = make(prob, 0)
    sd, td, rd, yd, ypd = dev.solve_schedule(touts, outputs=True)
    sh, th, rh, yh, yph = host.solve_schedule(touts, outputs=True)
    assert (sd == 0).all() and np.array_equal(rd, rh) and (rd == len(touts)).all()
    assert np.array_equal(yd, yh) and np.array_equal(ypd, yph)
    same(state(dev), state(host))
    assert dev.total_rounds() == host.total_rounds()
    cs, sl = make(prob, 1)
    ys = np.full_like(yd, np.nan)
    for _ in range(1000):
        s, t, r, yo, ypo = sl.solve_schedule(touts, max_rounds=37, outputs=True)
        m = ~np.isnan(yo)
        ys[m] = yo[m]
        if (s != 99).all():
            break
    assert (s == 0).all() and np.array_equal(ys, yd)
    same(state(sl), state(dev))


def test_stream_mode_on_the_device():
    """idaens_stream (finished systems restart at once, staggered first starts): same totals and states as the host stepper."""
    from idahip import problems
    prob = problems.lorenz63(batch=64)
    touts = prob["touts"][:10]
    cd, dev = make(prob, 1)
    ch, host = make(prob, 0)
    for k, stag in ((150, 40), (1, 0), (1, 0), (77, 0)):
        pd = dev.stream(touts, k, stagger_rounds=stag)
        ph = host.stream(touts, k, stagger_rounds=stag)
        assert pd == ph
        assert dev.total_rounds() == host.total_rounds() and dev.total_newton_iters() == host.total_newton_iters()
        same(state(dev), state(host))
    assert pd > 0


def test_failures_are_reported_like_the_host_stepper():
    """mxstep exhausted (recoverable TOO_MUCH_WORK, the next call continues) on both steppers."""
    from idahip import problems
    prob = problems.lorenz63(batch=16)
    cd, dev = make(prob, 1)
    ch, host = make(prob, 0)
    for e in (dev, host):
        e.set_max_num_steps(25)
    for _ in range(4):
        sd, td = dev.solve(2.0)
        sh, th = host.solve(2.0)
        assert np.array_equal(sd, sh) and np.array_equal(td, th)
        same(state(dev), state(host))
    assert (sd == -1).any() or (sd == 0).all()


@pytest.mark.parametrize("n,batch", [(24, 8), (64, 24), (200, 6), (704, 5), (1024, 3)])
def test_device_lock_step_rounds_equal_host_stepper_and_oracle(n, batch):
    """Linear dense problems, 8 < n <= 1024 (above 512 rows the leading super-panels of the LU take the workgroup-per-matrix
    panel kernels, with the list's length on the device like everywhere else): the rounds are enqueued from the host but decided on the device
    (idahip_round_solve). Per system the same steps as the host stepper and the oracle; the number of rounds may differ (a
    Newton solve that starts over with a fresh Jacobian does so in the next round)."""
    from idahip import problems
    prob = problems.linear_dense(n=n, batch=batch, procs=1)
    touts = prob["touts"]
    cd, dev = make(prob, 1)
    ch, host = make(prob, 0)
    for t in touts:
        sd, td = dev.solve(t)
        sh, th = host.solve(t)
        assert (sd == 0).all() and np.array_equal(sd, sh) and np.array_equal(td, th)
        same(state(dev), state(host))
    ref = run_oracle(prob, touts)
    c = dev.counters()
    for k in CNT:
        assert np.array_equal(c[k], ref["counters"][k]), k
    assert np.array_equal(dev.yy(), ref["yy"][-1]) and np.array_equal(dev.yp(), ref["yp"][-1])
    assert dev.total_rounds() >= host.total_rounds()


@pytest.mark.parametrize("n,batch,ntout", [(1536, 2, 2), (2048, 2, 1)])
def test_device_lock_step_rounds_for_linear_dense_beyond_1024_rows(n, batch, ntout):
    """SURVEY 8(f)-2, last piece: linear dense problems with 1024 < n <= 4096 on the device lock-step stepper too (the leading
    super-panels of their LU take the workgroup-per-matrix panel kernels, as the heat problem's do; the LU list's length is read
    back to size those launches). Per system the same steps as the host stepper and the oracle."""
    from idahip import problems
    prob = problems.linear_dense(n=n, batch=batch, procs=1)
    touts = [float(t) for t in prob["touts"][:ntout]]
    cd, dev = make(prob, 1)
    assert dev.device_controller_active() == 2
    ch, host = make(prob, 0)
    for t in touts:
        sd, td = dev.solve(t)
        sh, th = host.solve(t)
        assert (sd == 0).all() and np.array_equal(sd, sh) and np.array_equal(td, th)
        same(state(dev), state(host))
    ref = run_oracle(prob, touts)
    c = dev.counters()
    for k in CNT:
        assert np.array_equal(c[k], ref["counters"][k]), k
    assert np.array_equal(dev.yy(), ref["yy"][-1]) and np.array_equal(dev.yp(), ref["yp"][-1])


def test_device_lock_step_schedule_outputs_resume_and_stream():
    from idahip import problems
    prob = problems.linear_dense(n=48, batch=20, procs=1)
    touts = prob["touts"]
    cd, dev = make(prob, 1)
    ch, host = make(prob, 0)
    sd, td, rd, yd, ypd = dev.solve_schedule(touts, outputs=True)
    sh, th, rh, yh, yph = host.solve_schedule(touts, outputs=True)
    assert (sd == 0).all() and np.array_equal(rd, rh) and np.array_equal(yd, yh) and np.array_equal(ypd, yph)
    same(state(dev), state(host))
    cs, sl = make(prob, 1)
    ys = np.full_like(yd, np.nan)
    for _ in range(1000):
        s, t, r, yo, ypo = sl.solve_schedule(touts, max_rounds=23, outputs=True)
        m = ~np.isnan(yo)
        ys[m] = yo[m]
        if (s != 99).all():
            break
    assert (s == 0).all() and np.array_equal(ys, yd)
    same(state(sl), state(dev))
    # throughput mode: same totals and states as the host stepper as long as no Newton solve had to start over
    c2, d2 = make(prob, 1)
    c3, h2 = make(prob, 0)
    for k, stag in ((90, 30), (1, 0), (55, 0)):
        pd = d2.stream(touts, k, stagger_rounds=stag)
        ph = h2.stream(touts, k, stagger_rounds=stag)
        assert d2.total_rounds() == h2.total_rounds()
        if (h2.counter("nls_nconvfails") == 0).all() and h2.total_newton_iters() == d2.total_newton_iters():
            assert pd == ph
            same(state(d2), state(h2))
    assert pd > 0


@pytest.mark.parametrize("n,batch", [(48, 20), (200, 16)])
def test_switching_steppers_between_round_limited_calls(n, batch):
    """A round-limited call of the device lock-step stepper can leave a system INSIDE an attempt whose Newton solve has to
    start over with a linear setup in the next round (newton_retry, round_ida.hpp). The next call may run on the host stepper
    (idaens_set_device_controller(0), roots, tracing ...): it must continue that attempt -- not begin it again. One round per
    call, the stepper alternating from call to call, against the device stepper run alone.
    A linear problem never takes that path by itself (a stale Jacobian contracts at |1 - cjratio| / (1 + cjratio) <= 0.25), so
    the test makes the Jacobians stale the hard way: after thirty rounds -- the step sizes have settled by then and most steps
    reuse their factors; in the first rounds h doubles every step and every step sets up anew -- the user replaces A by 3 A
    (idahip_set_linear_dense between two solve calls). Every system whose next Newton solve starts on its old factors then diverges (rate > 0.9,
    ConvergenceRecover with jcur == false) -- in a device round, since call number thirty (counted from zero) is one -- and starts over in the call after
    it, which runs on the host stepper."""
    from idahip import problems
    prob = problems.linear_dense(n=n, batch=batch, procs=1)
    touts = prob["touts"]
    R0 = 30

    def swap(ctx):
        ctx.set_linear_dense(3.0 * prob["A"], prob["B"], prob["c"])

    cd, dev = make(prob, 1)
    yd = np.full((len(touts), batch, n), np.nan)
    s, t, r, yo, ypo = dev.solve_schedule(touts, max_rounds=R0, outputs=True)
    assert (s == 99).all(), "thirty rounds do not finish the schedule"
    m = ~np.isnan(yo)
    yd[m] = yo[m]
    assert int(dev.counter("nls_nconvfails").sum()) == 0
    swap(cd)
    for _ in range(4000):
        s, t, r, yo, ypo = dev.solve_schedule(touts, max_rounds=50, outputs=True)
        m = ~np.isnan(yo)
        yd[m] = yo[m]
        if (s != 99).all():
            break
    assert (s == 0).all()
    assert int(dev.counter("nls_nconvfails").sum()) > 0, "no Newton solve started over: the path under test was not taken"

    cm, mix = make(prob, 1)
    ym = np.full_like(yd, np.nan)
    handed_over = 0
    for i in range(8000):
        if i == R0:
            swap(cm)
        on_device = i % 2 == 0
        mix.set_device_controller(1 if on_device else 0)
        assert mix.device_controller_active() == (2 if on_device else 0)
        before = int(mix.counter("nls_nconvfails").sum())
        s, t, r, yo, ypo = mix.solve_schedule(touts, max_rounds=1, outputs=True)
        # a ConvergenceRecover on stale factors inside a DEVICE round is served in the next round, i.e. by the next call: host stepper
        if on_device and int(mix.counter("nls_nconvfails").sum()) > before:
            handed_over += 1
        m = ~np.isnan(yo)
        ym[m] = yo[m]
        if (s != 99).all():
            break
    assert (s == 0).all() and np.array_equal(ym, yd)
    same(state(mix), state(dev))
    assert handed_over > 0, "no one-round device call ended with a system waiting to start its Newton solve over"


@pytest.mark.parametrize("n,batch,ntout", [(40, 6, 10), (700, 3, 3), (1100, 3, 2), (4096, 4, 2)])
def test_device_lock_step_rounds_for_the_heat_problem(n, batch, ntout):
    """Config 4's problem on the device lock-step stepper (idahip_round_solve with the heat kernels; above 1024 rows the batched LU
    takes its workgroup-per-matrix panel kernels, the zero-block kernel and the helper workgroups with the list's length read on
    the device): per system the same steps as the host stepper and the oracle, every counter, at every output."""
    from idahip import problems
    prob = problems.heat1d(n=n, batch=batch)
    touts = [float(t) for t in prob["touts"][:ntout]]
    cd, dev = make(prob, 1)
    ch, host = make(prob, 0)
    ref = O.run_ensemble("heat1d", n, prob["yy0"], prob["yp0"], prob["rtol"], prob["atol"], touts, params=prob["params"], nthreads=batch)
    assert (ref["status"] == 0).all()
    for i, t in enumerate(touts):
        sd, td = dev.solve(t)
        sh, th = host.solve(t)
        assert (sd == 0).all() and np.array_equal(sd, sh) and np.array_equal(td, th)
        same(state(dev), state(host))
        assert np.array_equal(dev.yy(), ref["yy"][i]) and np.array_equal(dev.yp(), ref["yp"][i]), i
    c = dev.counters()
    for k in CNT:
        assert np.array_equal(c[k], ref["counters"][k]), k
    assert np.array_equal(c["kused"], ref["kused"]) and np.array_equal(dev.real("hused"), ref["hused"])
    assert dev.total_rounds() >= host.total_rounds() > 0


@pytest.mark.parametrize("kind,n,batch,ntout,period", [("heat1d", 1100, 6, 3, 3), ("heat1d", 40, 9, 10, 4), ("linear_dense", 200, 12, 10, 2),
                                                      ("linear_dense", 64, 16, 10, 5)])
def test_factorisations_batched_over_rounds_change_no_result(kind, n, batch, ntout, period):
    """idahip_set_lu_period(k): a lock-step round postpones its linear setups unless (k - 1) / k of the stepping systems ask for one
    (at most k - 1 rounds in a row); a system whose attempt calls for one waits, its attempt begun, while the others go on. Every system performs the same attempts with the same arithmetic: state, step
    sizes, orders and every counter at every output equal those of the unbatched run (and so the oracle's: the test above);
    only the number of rounds grows. Also across round-limited calls, which can leave a waiting system to the next call, and
    with the host stepper taking over a waiting system (it treats it like a Newton solve that starts over with a setup)."""
    from idahip import problems
    prob = problems.heat1d(n=n, batch=batch) if kind == "heat1d" else problems.linear_dense(n=n, batch=batch, procs=1)
    touts = [float(t) for t in prob["touts"][:ntout]]
    c1, plain = make(prob, 1)
    ck, held = make(prob, 1)
    assert ck.lu_period() == 1
    ck.set_lu_period(period)
    assert ck.lu_period() == period
    for t in touts:
        s1, t1 = plain.solve(t)
        sk, tk = held.solve(t)
        assert (s1 == 0).all() and np.array_equal(s1, sk) and np.array_equal(t1, tk)
        same(state(plain), state(held))
    assert held.total_rounds() >= plain.total_rounds() > 0
    print("rounds: %d unbatched, %d with period %d" % (plain.total_rounds(), held.total_rounds(), period))
    # round-limited calls (a waiting system is handed to the next call), every third call on the host stepper
    cm, mix = make(prob, 1)
    cm.set_lu_period(period)
    ym = np.full((len(touts), batch, n), np.nan)
    for i in range(20000):
        on_device = i % 3 != 2
        mix.set_device_controller(1 if on_device else 0)
        s, t, r, yo, ypo = mix.solve_schedule(touts, max_rounds=2, outputs=True)
        m = ~np.isnan(yo)
        ym[m] = yo[m]
        if (s != 99).all():
            break
    assert (s == 0).all()
    same(state(mix), state(plain))
    c2, sched = make(prob, 1)
    s, t, r, yo, ypo = sched.solve_schedule(touts, outputs=True)
    assert (s == 0).all() and np.array_equal(ym, yo)


@pytest.mark.parametrize("variant,superpanel", [(4, 1), (3, 1), (4, 0), (3, 0)])
def test_heat_setups_on_a_work_matrix_the_factorisation_left_zeroed(variant, superpanel, monkeypatch):
    """n >= 2048: with super-panel launches a factorisation leaves the ctx's work matrix all +0.0 (LuWs::jwzero) and the next
    Jacobian of the heat problem writes its band only; with the panel-by-panel pipeline it does not and the Jacobian kernel
    writes the whole matrix. Fourteen setups per system at n = 2120 on every pipeline combination, against the oracle."""
    import idahip
    from idahip import problems
    monkeypatch.setenv("IDAHIP_LU_SUPERPANEL", str(superpanel))
    prob = problems.heat1d(n=2120, batch=3)
    touts = [float(t) for t in prob["touts"][:3]]
    ctx = problems.make_ctx(prob)
    ctx.set_lu_variant(variant)
    assert ctx.lu_superpanel() == superpanel
    ens = idahip.Ensemble(ctx, prob["yy0"], prob["yp0"])
    ref = O.run_ensemble("heat1d", 2120, prob["yy0"], prob["yp0"], prob["rtol"], prob["atol"], touts, params=prob["params"], nthreads=3)
    for i, t in enumerate(touts):
        s, tr = ens.solve(t)
        assert (s == 0).all()
        assert np.array_equal(ens.yy(), ref["yy"][i]) and np.array_equal(ens.yp(), ref["yp"][i]), i
    c = ens.counters()
    for k in CNT:
        assert np.array_equal(c[k], ref["counters"][k]), k
    assert int(c["nsetups"].min()) >= 10
    ens.close()
    ctx.close()


def test_heat_stream_beyond_1024_rows_on_both_steppers():
    """Throughput mode for n > 1024 on the device lock-step stepper: there the round's LU-list length travels to the host (behind
    the residual kernels) to size the factorisation's launches, also when the rounds are enqueued without any other
    synchronisation. Same totals and states as the host stepper after the same rounds."""
    from idahip import problems
    prob = problems.heat1d(n=1100, batch=5)
    touts = prob["touts"]
    cd, dev = make(prob, 1)
    ch, host = make(prob, 0)
    for k, stag in ((9, 4), (1, 0), (7, 0)):
        pd = dev.stream(touts, k, stagger_rounds=stag)
        ph = host.stream(touts, k, stagger_rounds=stag)
        assert dev.total_rounds() == host.total_rounds() and dev.total_newton_iters() == host.total_newton_iters() and pd == ph
        same(state(dev), state(host))
    assert dev.total_newton_iters() > 0


def _run_with_roots(ens, touts, comps, thr, max_returns=400):
    """Ida::solve until every tout is reached, every return recorded (root returns do not advance the tout)."""
    ens.set_roots(comps, thr)
    rec = []
    for t in touts:
        for _ in range(max_returns):
            st, tret = ens.solve(float(t))
            rec.append((st.copy(), tret.copy(), ens.yy(), ens.yp(), ens.roots_found().copy()))
            assert (st >= 0).all()
            if (st == 0).all():
                break
        else:
            raise AssertionError("no end of root returns")
    return rec


@pytest.mark.parametrize("name", ["roberts", "linear_dense"])
def test_root_finding_on_the_device_steppers(name):
    """impl_r_check.rs on the device (ida_flow.hpp): the bracketing of idaens_set_roots' function family runs inside the
    one-thread-per-system stepper (Roberts: the reference example's two functions) and inside the lock-step rounds (linear
    dense, n = 24: two components crossing half of their final values). Every return -- status, t_ret, y, y', rootsfound --
    and the counters incl. the root-function evaluations equal the host stepper's, which the oracle pins
    (tests/test_gpu_ensemble.py::test_roberts_example_with_root_finding)."""
    from idahip import problems
    if name == "roberts":
        prob = roberts_batch(6)
        touts = [0.4, 4.0, 40.0]
        comps, thr = [0, 2], [0.97, 0.01]
    else:
        prob = problems.linear_dense(n=24, batch=7, procs=1)
        touts = [float(t) for t in prob["touts"][:4]]
        c0, plain = make(prob, 0)
        for t in touts:
            plain.solve(t)
        yend = plain.yy()
        comps = [1, 4]
        thr = [float(np.median(yend[:, 1])) * 0.5, float(np.median(yend[:, 4])) * 0.5]
    cd, dev = make(prob, 1)
    ch, host = make(prob, 0)
    rd = _run_with_roots(dev, touts, comps, thr)
    rh = _run_with_roots(host, touts, comps, thr)
    # with the roots set, the device ensemble still runs on its device stepper (the bracketing of ida_flow.hpp, not the host's)
    assert dev.device_controller_active() == (1 if name == "roberts" else 2) and host.device_controller_active() == 0
    assert len(rd) == len(rh) and any((r[0] == 2).any() for r in rh), "no root return in this run"
    for a, b in zip(rd, rh):
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
    same(state(dev), state(host))
    assert np.array_equal(dev.counter("nge"), host.counter("nge")) and (dev.counter("nge") > 0).all()
