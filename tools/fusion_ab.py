#!/usr/bin/env python3
"""Development tool: whole-pass rate (every system from fresh state through its output schedule) with the first two Newton
iterations fused into one device call (idahip_newton_iter2) against one host round trip per iteration, same work, alternating.
usage: python tools/fusion_ab.py [lorenz63|linear_dense|heat1d]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "rust-ida_amd"))
import numpy as np


def main():
    import idahip
    from idahip import problems
    wl = sys.argv[1] if len(sys.argv) > 1 else "lorenz63"
    prob = {"lorenz63": lambda: problems.lorenz63(batch=1024), "heat1d": lambda: problems.heat1d(n=4096, batch=256),
            "linear_dense": lambda: problems.linear_dense(n=512, batch=4096, procs=16)}[wl]()
    ctx = problems.make_ctx(prob)
    res = {0: [], 1: []}
    for rep in range(4):
        for fused in (1, 0):
            ens = idahip.Ensemble(ctx, prob["yy0"], prob["yp0"])
            ens.set_fused_newton(fused)
            ctx._chk(ctx.H.idahip_sync(ctx.h), "sync")
            t0 = time.perf_counter()
            status, _, reached = ens.solve_schedule(prob["touts"])
            ctx._chk(ctx.H.idahip_sync(ctx.h), "sync")
            dt = time.perf_counter() - t0
            assert (status == 0).all()
            if rep > 0:
                res[fused].append(ens.total_newton_iters() / dt)
            ens.close()
    for fused in (1, 0):
        print("%s: fused=%d  %.0f Newton iters/s (median of %d whole passes)" % (wl, fused, float(np.median(res[fused])), len(res[fused])))


if __name__ == "__main__":
    main()
