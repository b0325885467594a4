#!/usr/bin/env python3
"""Development tool (round 5): SURVEY 8(d)'s whole pass (every system from fresh state through its schedule) with the rank's systems
as G groups side by side -- does starting group g a little after group g - 1 help? In a whole pass every group is in the same phase
of its integrations (all systems set up a Jacobian in the first rounds), so the groups' LU-heavy and HBM-heavy rounds coincide.
usage: python tools/whole_pass_offsets.py [B] [G]"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rust-ida_amd"))
import numpy as np
import idahip
from idahip import problems


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    G = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    n = 512
    procs = int(os.environ.get("IDAHIP_GEN_PROCS", "16"))
    if procs > 1:
        problems.ensure_fork_server()
    full = problems.linear_dense(n=n, batch=B, procs=procs)
    per = B // G
    streams, nconc = idahip.concurrent_streams(G)
    subs = [{k: (v[g * per:(g + 1) * per] if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[0] == B else v) for k, v in full.items()} for g in range(G)]
    ctxs = [problems.make_ctx(s, stream=streams[g]) for g, s in enumerate(subs)]
    touts = full["touts"]
    for off_ms in (0.0, 0.0, 5.0, 10.0, 20.0, 40.0, 80.0, 0.0):
        enss = [idahip.Ensemble(c, s["yy0"], s["yp0"]) for c, s in zip(ctxs, subs)]
        for c in ctxs:
            c._chk(c.H.idahip_sync(c.h), "sync")

        def go(e, d):
            if d > 0:
                time.sleep(d * 1e-3)
            st, _, re_ = e.solve_schedule(touts)
            assert (st == 0).all()
        th = [threading.Thread(target=go, args=(enss[g], g * off_ms)) for g in range(G)]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        dt = time.perf_counter() - t0
        it = sum(e.total_newton_iters() for e in enss)
        print("%d groups, group g starts %.0f ms after group g - 1: %.3f s, %.1f k iters/s" % (G, off_ms, dt, it / dt / 1e3), flush=True)
        for e in enss:
            e.close()


if __name__ == "__main__":
    main()
