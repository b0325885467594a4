#!/usr/bin/env python3
"""Development tool: do the batched LU (fp64 VALU bound) and the residual / Newton-iteration kernels (HBM bound) of OTHER
systems overlap when issued on two streams? Two contexts on one device, driven from two host threads.
usage: python tools/overlap_probe.py [nlu] [nother]"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rust-ida_amd"))
import numpy as np
import idahip
from idahip import problems


def main():
    nlu = int(sys.argv[1]) if len(sys.argv) > 1 else 1240
    nother = int(sys.argv[2]) if len(sys.argv) > 2 else 2856
    n = 512
    p1 = problems.linear_dense(n=n, batch=nlu, procs=16)
    p2 = problems.linear_dense(n=n, batch=nother, first=8192, procs=16)
    c1, c2 = problems.make_ctx(p1), problems.make_ctx(p2)
    for c, p in ((c1, p1), (c2, p2)):
        c.upload(idahip.F_YY, p["yy0"]); c.upload(idahip.F_YP, p["yp0"])
        c.upload(idahip.F_YYPREDICT, p["yy0"]); c.upload(idahip.F_YPPREDICT, p["yp0"])
        c.upload(idahip.F_EWT, np.ones_like(p["yy0"]))
    c2.nls_lsetup(0.0, 100.0)  # factors for the Newton iterations of the "other" systems

    def lu():
        c1.nls_lsetup(0.0, 100.0)

    def other(reps=3):
        for _ in range(reps):   # ~ the residual + Newton-iteration work of the systems that need no setup in a round
            c2.nls_sys(0.0, 100.0, True)
            c2.newton_iter(np.ones(nother))

    def timed(f):
        t0 = time.perf_counter(); f(); return (time.perf_counter() - t0) * 1e3

    lu(); other()
    t_lu = min(timed(lu) for _ in range(3))
    t_ot = min(timed(other) for _ in range(3))

    def both():
        a = threading.Thread(target=lu); b = threading.Thread(target=other)
        a.start(); b.start(); a.join(); b.join()
    t_both = min(timed(both) for _ in range(4))
    print("LU of %d matrices alone %.2f ms | sys+newton x3 of %d systems alone %.2f ms | sum %.2f | concurrent %.2f ms" %
          (nlu, t_lu, nother, t_ot, t_lu + t_ot, t_both))


if __name__ == "__main__":
    main()
