"""Parity at BASELINE.json's full single-GPU size (config 3: N = 512, B = 4096 systems resident on one device).

The oracle cannot integrate 4096 systems of N = 512 in test time, so the full-size run is checked through properties that
do not depend on the batch size, plus the oracle itself on a sample:
  * a sample of systems spread over the batch is bit-identical (state, step sizes, orders, every counter) to the
    oracle integrating those systems alone, over the whole horizon t = 0 .. 1 (all ten outputs of config 3);
  * batch-position independence: the same systems integrated as a small batch of their own give the same bits as
    inside the full batch (no cross-talk between workgroups, index lists, staging rings);
  * the batched LU of all 4096 Jacobians satisfies P J = L U to rounding on sampled systems and its forward/back
    substitution solves J x = b; the factors of the sampled systems are bit-identical to the oracle's.
Input generation (17 GB) dominates the run time; everything shares one module-scoped fixture.
"""
import os

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
N, B = 512, 4096
SAMPLE = np.array([0, 1, 63, 64, 777, 1023, 1024, 2047, 2048, 3000, 4094, 4095])
CNT = ("nst", "nre", "nje", "nsetups", "nni", "netf", "ncfn", "n_attempts")


@pytest.fixture(scope="module")
def full():
    import idahip
    from idahip import problems
    # numpy worker processes of a fork server (nothing is forked from this process, whose ROCm runtime is live by now)
    procs = max(1, min(32, (os.cpu_count() or 1) // 2))
    prob = problems.linear_dense(n=N, batch=B, procs=procs)
    ctx = problems.make_ctx(prob)
    ens = idahip.Ensemble(ctx, prob["yy0"], prob["yp0"])
    touts = [float(t) for t in prob["touts"]]  # the ten Ida::solve calls of config 3: t = 0.1 .. 1.0
    for tout in touts:
        status, tret = ens.solve(tout)
        assert (status == 0).all()
        assert np.array_equal(tret, np.full(B, tout))
    out = {"prob": prob, "ctx": ctx, "ens": ens, "touts": touts, "tout": touts[-1], "yy": ens.yy(), "yp": ens.yp(),
           "counters": ens.counters(), "hused": ens.real("hused")}
    yield out
    ens.close()


def sub_problem(prob, ids):
    return {k: (v[ids] if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[0] == B else v) for k, v in prob.items()}


def test_sampled_systems_match_the_oracle_bit_for_bit(full):
    sub = sub_problem(full["prob"], SAMPLE)
    ref = O.run_ensemble("linear_dense", N, sub["yy0"], sub["yp0"], sub["rtol"], sub["atol"], full["touts"],
                         A=sub["A"], B=sub["B"], c=sub["c"], nthreads=len(SAMPLE))
    assert (ref["status"] == 0).all()
    for k in CNT:
        assert np.array_equal(full["counters"][k][SAMPLE], ref["counters"][k]), k
    assert np.array_equal(full["counters"]["kused"][SAMPLE], ref["kused"])
    assert np.array_equal(full["hused"][SAMPLE], ref["hused"])
    assert np.array_equal(full["yy"][SAMPLE], ref["yy"][-1])
    assert np.array_equal(full["yp"][SAMPLE], ref["yp"][-1])


def test_every_system_of_the_batch_matches_the_oracle_bit_for_bit(full):
    """Exhaustive, not sampled: the oracle integrates ALL 4096 systems of config 3 over the whole horizon on the box's host
    cores (one std::thread per core, about 40 s on 64 cores) and every system's final state, last step size and order and
    every counter must equal the device's. (On a box with fewer than 32 cores: the first 512 systems.)"""
    ncpu = os.cpu_count() or 1
    count = B if ncpu >= 32 else 512
    prob = full["prob"]
    ref = O.run_ensemble("linear_dense", N, prob["yy0"][:count], prob["yp0"][:count], prob["rtol"], prob["atol"], full["touts"],
                         A=prob["A"][:count], B=prob["B"][:count], c=prob["c"][:count], nthreads=min(ncpu, 64))
    assert (ref["status"] == 0).all()
    for k in CNT:
        assert np.array_equal(full["counters"][k][:count], ref["counters"][k]), k
    assert np.array_equal(full["counters"]["kused"][:count], ref["kused"])
    assert np.array_equal(full["hused"][:count], ref["hused"])
    assert np.array_equal(full["yy"][:count], ref["yy"][-1])
    assert np.array_equal(full["yp"][:count], ref["yp"][-1])
    print("oracle: %d systems in %.1f s on %d threads" % (count, ref["seconds"], min(ncpu, 64)))


QUIRK_PATHS = ("ncfn", "nlufail", "nconv_jcur", "nfail_first")


def test_config3_stays_clear_of_the_paths_where_the_oracle_follows_c_ida(full):
    """SURVEY.md 9 / DESIGN.md 2: on a zero pivot (Q2), when Newton gives up with a current Jacobian (Q3/Q4) and on a failed
    attempt before the first step (Q5) oracle and product do what C IDA does, not what the reference's text does. None of
    the 4096 systems of config 3 takes any of those paths over the whole horizon, nor does a Newton solve start over with a
    fresh Jacobian: "identical to the reference" is claimed only where the reference's text and C IDA agree."""
    c = full["counters"]
    for k in QUIRK_PATHS + ("nls_nconvfails",):
        assert int(c[k].sum()) == 0, k
    assert int(c["nge"].sum()) == 0  # no root functions are set


def test_result_does_not_depend_on_the_position_in_the_batch(full):
    import idahip
    from idahip import problems
    ids = np.arange(1500, 1500 + 48)
    sub = sub_problem(full["prob"], ids)
    ctx = problems.make_ctx(sub)
    ens = idahip.Ensemble(ctx, sub["yy0"], sub["yp0"])
    for tout in full["touts"]:
        status, _ = ens.solve(tout)
        assert (status == 0).all()
    c = ens.counters()
    for k in CNT:
        assert np.array_equal(c[k], full["counters"][k][ids]), k
    assert np.array_equal(ens.yy(), full["yy"][ids])
    assert np.array_equal(ens.yp(), full["yp"][ids])
    assert np.array_equal(ens.real("hused"), full["hused"][ids])
    ens.close()


def test_full_batch_lu_factors_and_solves(full):
    """idahip_nls_lsetup on all 4096 systems at once (J = B + cj A), then P J = L U and J x = b on sampled systems."""
    import idahip
    ctx, prob = full["ctx"], full["prob"]
    cj = 37.5
    rc, info = ctx.nls_lsetup(full["tout"], cj)
    assert rc == 0 and not info.any()
    rng = np.random.default_rng(5)
    for s in SAMPLE[::3]:
        lu, piv = ctx.download_lu(int(s))
        J = (prob["B"][s] + cj * prob["A"][s]).T                  # stored [col][row] -> logical [row][col]
        info_o, lu_o, piv_o = O.getrf(J)
        assert info_o == 0
        assert np.array_equal(piv, piv_o)
        assert np.array_equal(lu, lu_o)
        L = np.tril(lu, -1) + np.eye(N)
        U = np.triu(lu)
        PJ = J.copy()
        for k in range(N):                                         # the reference's row interchanges, dense.rs:125-131
            l = int(piv[k])
            if l != k:
                PJ[[k, l]] = PJ[[l, k]]
        assert np.abs(L @ U - PJ).max() <= 1e-11 * np.abs(J).max() * N
        b = rng.standard_normal(N)
        x = O.getrs(lu, piv, b)
        assert np.abs(J @ x - b).max() <= 1e-11 * N * (np.abs(J) @ np.abs(x)).max()  # backward-stable solve


@pytest.mark.parametrize("shard", [3, 7])
def test_config5_shard_inputs_match_the_oracle(shard):
    """Config 5 = config 3's generator over 32,768 systems, shard s = systems [4096 s, 4096 (s + 1)). Its inputs differ
    from config 3's only through the per-system seed, so a sample of a late shard (first, middle, last systems of shards
    3 and 7), integrated over the whole horizon on the HIP path, must equal the oracle bit for bit like shard 0 does."""
    import idahip
    from idahip import problems
    first = 4096 * shard
    offs = np.array([0, 1, 2047, 2048, 4094, 4095])
    parts = [problems.linear_dense(n=N, batch=1, first=first + int(o)) for o in offs]
    p = dict(parts[0])
    for k in ("A", "B", "c", "yy0", "yp0"):
        p[k] = np.concatenate([q[k] for q in parts])
    touts = [float(t) for t in p["touts"]]
    ctx = problems.make_ctx(p)
    ens = idahip.Ensemble(ctx, p["yy0"], p["yp0"])
    status, tret, reached, yo, ypo = ens.solve_schedule(touts, outputs=True)
    assert (status == 0).all() and (reached == len(touts)).all()
    ref = O.run_ensemble("linear_dense", N, p["yy0"], p["yp0"], p["rtol"], p["atol"], touts, A=p["A"], B=p["B"], c=p["c"],
                         nthreads=len(offs))
    assert (ref["status"] == 0).all()
    c = ens.counters()
    for k in CNT:
        assert np.array_equal(c[k], ref["counters"][k]), k
    assert np.array_equal(yo, ref["yy"]) and np.array_equal(ypo, ref["yp"])
    assert np.array_equal(ens.real("hused"), ref["hused"])
    ens.close()


def test_four_groups_side_by_side_equal_the_single_ensemble_at_full_size(full):
    """The headline configuration as bench.py runs it since round 5 -- the 4096 systems as four ensembles of 1024 side by side, each
    on its own context, probed-concurrent HIP stream and host thread (idaens_solve_schedule_group) -- against the single ensemble of
    the fixture (which the oracle pins on its sample): every system's state, step size, order and counters, bit for bit."""
    import idahip
    from idahip import problems
    prob = full["prob"]
    G, per = 4, B // 4
    streams, nconc = idahip.concurrent_streams(G)
    assert nconc >= 2, "the device offered no two concurrent hardware queues"
    subs = [sub_problem(prob, np.arange(g * per, (g + 1) * per)) for g in range(G)]
    ctxs = [problems.make_ctx(s_, stream=streams[g]) for g, s_ in enumerate(subs)]
    enss = [idahip.Ensemble(c, s_["yy0"], s_["yp0"]) for c, s_ in zip(ctxs, subs)]
    res = idahip.solve_schedule_group(enss, full["touts"])
    for status, tret, reached in res:
        assert (status == 0).all() and (reached == len(full["touts"])).all() and np.array_equal(tret, np.full(per, full["tout"]))
    cs = [e.counters() for e in enss]
    for k in CNT + ("kused",):
        assert np.array_equal(np.concatenate([c[k] for c in cs]), full["counters"][k]), k
    assert np.array_equal(np.concatenate([e.yy() for e in enss]), full["yy"])
    assert np.array_equal(np.concatenate([e.yp() for e in enss]), full["yp"])
    assert np.array_equal(np.concatenate([e.real("hused") for e in enss]), full["hused"])
    for e in enss:
        e.close()
    for c in ctxs:
        c.close()
    idahip.release_streams(streams)


def test_stream_driver_at_full_n(full):
    """idaens_stream (the driver bench.py times) at N = 512: systems restart from their initial conditions when they reach
    t = 1; whatever pass a system is in, its accepted steps are those of the oracle's fresh integration."""
    import idahip
    from idahip import problems
    ids = np.array([5, 1500, 4000, 4095])
    sub = sub_problem(full["prob"], ids)
    ens = idahip.Ensemble(problems.make_ctx(sub), sub["yy0"], sub["yp0"])
    steps = []
    for s in range(len(ids)):
        o = O.OracleIda("linear_dense", N, sub["yy0"][s], sub["yp0"][s], sub["rtol"], sub["atol"], A=sub["A"][s], B=sub["B"][s],
                        c=sub["c"][s])
        O.lib().oracle_ida_record_steps(o.h, 1)
        for t in sub["touts"]:
            assert o.solve(float(t))[0] == 0
        steps.append(o.recorded_steps())
    passes = 0
    while passes < 2 * len(ids):
        passes = ens.stream(sub["touts"], 5, stagger_rounds=3)
        nst, tn, hu, ku = ens.counter("nst"), ens.real("tn"), ens.real("hused"), ens.counter("kused")
        for s in range(len(ids)):
            if nst[s] > 0:
                assert np.array_equal(steps[s][nst[s] - 1, :3], [tn[s], hu[s], float(ku[s])]), (s, nst[s])
    ens.close()


def test_config2_lorenz63_full_batch():
    """Config 2 at its full size: 65,536 Lorenz systems (N = 3, the one-thread-per-system kernels), every system
    bit-identical to the oracle -- state and counters -- at t = 0.1 .. 2.0."""
    import idahip
    from idahip import problems
    p = problems.lorenz63(batch=65536)
    touts = p["touts"][:20]
    ctx = problems.make_ctx(p)
    ens = idahip.Ensemble(ctx, p["yy0"], p["yp0"])
    for t in touts:
        status, _ = ens.solve(float(t))
        assert (status == 0).all()
    ref = O.run_ensemble("lorenz63", 3, p["yy0"], p["yp0"], p["rtol"], p["atol"], touts, params=p["params"],
                         nthreads=min(64, os.cpu_count() or 1))
    assert (ref["status"] == 0).all()
    c = ens.counters()
    for k in CNT:
        assert np.array_equal(c[k], ref["counters"][k]), k
    assert np.array_equal(ens.yy(), ref["yy"][-1]) and np.array_equal(ens.yp(), ref["yp"][-1])
    assert np.array_equal(ens.real("hused"), ref["hused"])
    for k in QUIRK_PATHS:  # config 2 stays clear of the C-IDA paths (Newton-internal re-setups are the reference's own path)
        assert int(c[k].sum()) == 0, k
    ens.close()


@pytest.mark.parametrize("device_ctl", [1, 0])
def test_config2_lorenz63_whole_horizon(device_ctl):
    """Config 2 exactly as BASELINE.json states it -- Lorenz63, N = 3, B = 1024 -- over its WHOLE horizon: all 50 outputs
    t = 0.1 .. 5.0 against the oracle, y and y' at every output and every counter, step size and order at the end (round 4
    compared the first 20 outputs only; the late horizon is where a slip in the controller -- a step size, an order -- would have
    had the longest time to show). Device stepper (one thread per system) and host stepper."""
    import idahip
    from idahip import problems
    p = problems.lorenz63(batch=1024)
    touts = p["touts"]
    assert len(touts) == 50 and abs(float(touts[-1]) - 5.0) < 1e-12
    ctx = problems.make_ctx(p)
    ens = idahip.Ensemble(ctx, p["yy0"], p["yp0"])
    ens.set_device_controller(device_ctl)
    assert ens.device_controller_active() == (1 if device_ctl else 0)
    status, tret, reached, yo, ypo = ens.solve_schedule(touts, outputs=True)
    assert (status == 0).all() and (reached == 50).all() and np.array_equal(tret, np.full(1024, touts[-1]))
    ref = O.run_ensemble("lorenz63", 3, p["yy0"], p["yp0"], p["rtol"], p["atol"], touts, params=p["params"],
                         nthreads=min(64, os.cpu_count() or 1))
    assert (ref["status"] == 0).all()
    for i in range(50):
        assert np.array_equal(yo[i], ref["yy"][i]) and np.array_equal(ypo[i], ref["yp"][i]), "output %d (t = %g)" % (i, touts[i])
    c = ens.counters()
    for k in CNT:
        assert np.array_equal(c[k], ref["counters"][k]), k
    assert np.array_equal(c["kused"], ref["kused"]) and np.array_equal(ens.real("hused"), ref["hused"])
    assert np.array_equal(ens.yy(), ref["yy"][-1]) and np.array_equal(ens.yp(), ref["yp"][-1])
    for k in QUIRK_PATHS:
        assert int(c[k].sum()) == 0, k
    ens.close()
    ctx.close()


@pytest.mark.parametrize("n,batch,ntout", [(1024, 4, 2), (4096, 256, 1)])
def test_config4_heat1d(n, batch, ntout):
    """Config 4 (method-of-lines heat equation, tridiagonal Jacobian: the a_kj == 0 paths of the LU everywhere) at
    N = 1024 (largest N of the two-rows-per-lane panels) and at its full size, N = 4096 with 256 systems (eight rows
    per lane in the leading super-panels)."""
    import idahip
    from idahip import problems
    p = problems.heat1d(n=n, batch=batch)
    touts = p["touts"][:ntout]
    ctx = problems.make_ctx(p)
    ens = idahip.Ensemble(ctx, p["yy0"], p["yp0"])
    for t in touts:
        status, _ = ens.solve(float(t))
        assert (status == 0).all()
    ref = O.run_ensemble("heat1d", n, p["yy0"], p["yp0"], p["rtol"], p["atol"], touts, params=p["params"],
                         nthreads=min(batch, 64, os.cpu_count() or 1))
    assert (ref["status"] == 0).all()
    c = ens.counters()
    for k in CNT:
        assert np.array_equal(c[k], ref["counters"][k]), k
    assert np.array_equal(ens.yy(), ref["yy"][-1]) and np.array_equal(ens.yp(), ref["yp"][-1])
    for k in QUIRK_PATHS + ("nls_nconvfails",):  # config 4 stays clear of the C-IDA paths
        assert int(c[k].sum()) == 0, k
    ens.close()


def test_config4_heat1d_whole_horizon():
    """Config 4 over its WHOLE horizon (all ten outputs, t = 0.01 .. 0.1) at N = 4096, ALL 256 systems of the batch (on a box with
    fewer than 32 host cores: six systems sampled from it; kappa_b = 1 + b / 256 depends on the global id): state, step sizes,
    orders and counters bit-identical to the oracle at every output. (The oracle needs about 4.5 s per system and core.)"""
    import idahip
    from idahip import problems
    full = problems.heat1d(n=4096, batch=256)
    ids = np.arange(256) if (os.cpu_count() or 1) >= 32 else np.array([0, 37, 101, 128, 200, 255])
    p = {k: (v[ids] if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[0] == 256 else v) for k, v in full.items()}
    touts = [float(t) for t in p["touts"]]
    assert len(touts) == 10
    ctx = problems.make_ctx(p)
    ens = idahip.Ensemble(ctx, p["yy0"], p["yp0"])
    ref = O.run_ensemble("heat1d", 4096, p["yy0"], p["yp0"], p["rtol"], p["atol"], touts, params=p["params"],
                         nthreads=min(len(ids), 64, os.cpu_count() or 1))
    assert (ref["status"] == 0).all()
    print("oracle: %d systems of N = 4096 in %.1f s" % (len(ids), ref["seconds"]))
    for i, t in enumerate(touts):
        status, tret = ens.solve(t)
        assert (status == 0).all() and np.array_equal(tret, np.full(len(ids), t))
        assert np.array_equal(ens.yy(), ref["yy"][i]) and np.array_equal(ens.yp(), ref["yp"][i]), i
    c = ens.counters()
    for k in CNT:
        assert np.array_equal(c[k], ref["counters"][k]), k
    assert np.array_equal(c["kused"], ref["kused"]) and np.array_equal(ens.real("hused"), ref["hused"])
    for k in QUIRK_PATHS + ("nls_nconvfails",):
        assert int(c[k].sum()) == 0, k
    ens.close()

