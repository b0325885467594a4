"""idaens_stream_group / idaens_solve_schedule_group (include/ida_ensemble.h): several ensembles side by side on one device,
one host thread and one HIP stream each. The reference has no notion of a batch -- one `Ida` object per IVP
(/root/reference/src/lib.rs:89-244) -- so how the systems are grouped must not show in any system's result: a group run
equals the same systems integrated as one ensemble (and the oracle pins that one elsewhere), bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
CNT = ("nst", "nre", "nje", "nsetups", "nni", "netf", "ncfn", "n_attempts", "kused")


def _split(prob, lo, hi, total):
    return {k: (v[lo:hi] if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[0] == total else v) for k, v in prob.items()}


@pytest.mark.parametrize("kind,n,batch,cuts", [("linear_dense", 64, 40, (0, 13, 27, 40)), ("linear_dense", 200, 9, (0, 4, 9)), ("heat1d", 96, 12, (0, 6, 12))])
def test_schedule_of_a_group_equals_the_single_ensemble(kind, n, batch, cuts):
    import idahip
    from idahip import problems
    prob = problems.linear_dense(n=n, batch=batch, procs=1) if kind == "linear_dense" else problems.heat1d(n=n, batch=batch)
    touts = prob["touts"][:6]
    c1 = problems.make_ctx(prob)
    one = idahip.Ensemble(c1, prob["yy0"], prob["yp0"])
    st, tr, re_ = one.solve_schedule(touts)
    assert (st == 0).all() and (re_ == len(touts)).all()
    parts = [_split(prob, lo, hi, batch) for lo, hi in zip(cuts[:-1], cuts[1:])]
    ctxs = [problems.make_ctx(p) for p in parts]
    enss = [idahip.Ensemble(c, p["yy0"], p["yp0"]) for c, p in zip(ctxs, parts)]
    res = idahip.solve_schedule_group(enss, touts)
    assert np.array_equal(np.concatenate([r[0] for r in res]), st) and np.array_equal(np.concatenate([r[1] for r in res]), tr)
    assert np.array_equal(np.concatenate([r[2] for r in res]), re_)
    c_one = one.counters()
    cs = [e.counters() for e in enss]
    for k in CNT:
        assert np.array_equal(np.concatenate([c[k] for c in cs]), c_one[k]), k
    assert np.array_equal(np.concatenate([e.yy() for e in enss]), one.yy())
    assert np.array_equal(np.concatenate([e.yp() for e in enss]), one.yp())
    assert np.array_equal(np.concatenate([e.real("hused") for e in enss]), one.real("hused"))
    for e in enss + [one]:
        e.close()
    for c in ctxs + [c1]:
        c.close()


def test_stream_of_a_group_equals_its_members_run_alone():
    """Throughput mode: the same three ensembles streamed side by side (with a start offset) and one after the other end in the
    same per-system states, counters and totals after the same number of rounds."""
    import idahip
    from idahip import problems
    prob = problems.linear_dense(n=48, batch=30, procs=1)
    cuts = (0, 10, 20, 30)
    parts = [_split(prob, lo, hi, 30) for lo, hi in zip(cuts[:-1], cuts[1:])]

    def make():
        ctxs = [problems.make_ctx(p) for p in parts]
        return ctxs, [idahip.Ensemble(c, p["yy0"], p["yp0"]) for c, p in zip(ctxs, parts)]
    ca, side = make()
    cb, alone = make()
    for k, stag in ((70, 25), (1, 0), (33, 0)):
        done = idahip.stream_group(side, prob["touts"], k, stagger_rounds=stag, offset_us=300)
        for g, e in enumerate(alone):
            assert e.stream(prob["touts"], k, stagger_rounds=stag) == done[g]
        for a, b in zip(side, alone):
            assert a.total_rounds() == b.total_rounds() and a.total_newton_iters() == b.total_newton_iters()
            ca_, cb_ = a.counters(), b.counters()
            for kk in CNT:
                assert np.array_equal(ca_[kk], cb_[kk]), kk
            assert np.array_equal(a.yy(), b.yy()) and np.array_equal(a.real("tn"), b.real("tn"))
    assert sum(done) > 0
    for e in side + alone:
        e.close()
    for c in ca + cb:
        c.close()


def test_a_group_needs_a_context_per_ensemble():
    import idahip
    from idahip import problems
    prob = problems.linear_dense(n=24, batch=4, procs=1)
    ctx = problems.make_ctx(prob)
    a = idahip.Ensemble(ctx, prob["yy0"], prob["yp0"])
    with pytest.raises(idahip.IdaHipError):
        idahip.stream_group([a, a], prob["touts"], 2)
    a.close()
    ctx.close()


def test_streamed_contexts_for_overlapping_parts_hold_the_same_systems():
    """problems.make_ctxs_linear_dense_streamed (bench.py's input path): the systems generated once, slice by slice, and
    uploaded to every part that covers them -- three groups and one context with all of them. Integrating on those contexts
    gives what integrating problems.linear_dense's arrays gives."""
    import idahip
    from idahip import problems
    n, first, count = 32, 100, 23
    parts = [(0, 8, None), (8, 8, None), (16, 7, None), (0, 23, None)]
    made = problems.make_ctxs_linear_dense_streamed(n, first, count, parts, procs=1, keep=5, slice_bytes=16 * n * n * 5)  # five systems per slice
    ref = problems.linear_dense(n=n, batch=count, first=first, procs=1)
    c0 = problems.make_ctx(ref)
    e0 = idahip.Ensemble(c0, ref["yy0"], ref["yp0"])
    st, _, _ = e0.solve_schedule(ref["touts"][:3])
    assert (st == 0).all()
    assert np.array_equal(made[0][1]["A"], ref["A"][:5]) and made[1][1]["A"].shape[0] == 0 and np.array_equal(made[3][1]["B"], ref["B"][:5])
    for (ctx, prob), (off, cnt, _) in zip(made, parts):
        assert np.array_equal(prob["yy0"], ref["yy0"][off:off + cnt]) and np.array_equal(prob["c"], ref["c"][off:off + cnt])
        e = idahip.Ensemble(ctx, prob["yy0"], prob["yp0"])
        st, _, _ = e.solve_schedule(ref["touts"][:3])
        assert (st == 0).all()
        assert np.array_equal(e.yy(), e0.yy()[off:off + cnt]) and np.array_equal(e.counter("nni"), e0.counter("nni")[off:off + cnt])
        e.close()
        ctx.close()
    e0.close()
    c0.close()


def test_concurrent_streams_are_usable_and_distinct():
    import idahip
    streams, nconc = idahip.concurrent_streams(4)
    assert len(streams) == 4 and len({s.value for s in streams}) == 4 and 1 <= nconc <= 4
    from idahip import problems
    prob = problems.linear_dense(n=24, batch=4, procs=1)
    ctx = problems.make_ctx(prob, stream=streams[3])
    e = idahip.Ensemble(ctx, prob["yy0"], prob["yp0"])
    st, _ = e.solve(0.1)
    assert (st == 0).all()
    e.close()
    ctx.close()
    # the streams the probe calls concurrent also share the chip evenly when both want all of it; a stream with itself takes turns
    assert idahip.stream_pair_share(streams[0], streams[1]) > 0.8 if nconc >= 2 else True
    assert idahip.stream_pair_share(streams[0], streams[0]) < 0.2
    idahip.release_streams(streams)
