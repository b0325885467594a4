#!/usr/bin/env python3
"""Development tool: does the batched LU gain from two groups of matrices in flight on two streams (one group's ramps, tails
and latency-bound super-panel kernels filled by the other group's work)? Wall time of Jacobian + LU (idahip_nls_lsetup) for
2 x B/2 matrices on two contexts driven by two host threads, against B matrices on one. usage: python tools/two_stream_lu.py [B]"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rust-ida_amd"))
import numpy as np
import idahip
from idahip import problems


def mk(p, lo, hi):
    q = {k: (v[lo:hi] if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[0] == p["A"].shape[0] else v) for k, v in p.items()}
    ctx = problems.make_ctx(q)
    ctx.upload(idahip.F_YY, q["yy0"]); ctx.upload(idahip.F_YP, q["yp0"])
    return ctx


def main():
    n, B = 512, int(sys.argv[1]) if len(sys.argv) > 1 else 1240
    groups = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    p = problems.linear_dense(n=n, batch=B, procs=int(os.environ.get("IDAHIP_GEN_PROCS", "16")))
    one = mk(p, 0, B)
    parts = [mk(p, g * B // groups, (g + 1) * B // groups) for g in range(groups)]

    def run(c):
        c.nls_lsetup(0.0, 100.0)

    for c in [one] + parts:
        run(c)
    for rep in range(3):
        t0 = time.perf_counter(); run(one); t1 = time.perf_counter() - t0
        t0 = time.perf_counter()
        for c in parts:
            run(c)
        t2 = time.perf_counter() - t0
        th = [threading.Thread(target=run, args=(c,)) for c in parts]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        t3 = time.perf_counter() - t0
        print("rep %d: one ctx %.3f ms | %d groups one after the other %.3f ms | %d groups on %d threads/streams %.3f ms" % (rep, t1 * 1e3, groups, t2 * 1e3, groups, groups, t3 * 1e3))


if __name__ == "__main__":
    main()
