#!/usr/bin/env python3
"""Development tool: overlap_probe.py with stream PRIORITIES. The batched LU (fp64 VALU bound, fills every CU's registers
and LDS) on one stream, the residual / Newton-iteration kernels (HBM bound) of other systems on another stream created with
hipStreamCreateWithPriority: does the workgroup dispatcher let the high-priority stream's workgroups in between the LU's,
and is the sum shorter than the two alone?  usage: python tools/overlap_prio.py [nlu] [nother]"""
import ctypes as C, os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rust-ida_amd"))
import numpy as np
import idahip
from idahip import problems

hip = C.CDLL("libamdhip64.so")


def mkstream(prio):
    s = C.c_void_p()
    rc = hip.hipStreamCreateWithPriority(C.byref(s), C.c_uint(0), C.c_int(prio))
    assert rc == 0, rc
    return s


def main():
    nlu = int(sys.argv[1]) if len(sys.argv) > 1 else 1240
    nother = int(sys.argv[2]) if len(sys.argv) > 2 else 2856
    n = 512
    lo, hi = C.c_int(), C.c_int()
    hip.hipInit(0)
    hip.hipSetDevice(0)
    hip.hipDeviceGetStreamPriorityRange(C.byref(lo), C.byref(hi))
    print("priority range: least %d greatest %d" % (lo.value, hi.value), flush=True)
    p1 = problems.linear_dense(n=n, batch=nlu, procs=16)
    p2 = problems.linear_dense(n=n, batch=nother, first=8192, procs=16)
    for name, plu, pot in (("default/default", None, None), ("lu least / other greatest", lo.value, hi.value),
                           ("lu greatest / other least", hi.value, lo.value), ("both greatest", hi.value, hi.value)):
        s1 = mkstream(plu) if plu is not None else None
        s2 = mkstream(pot) if pot is not None else None
        c1, c2 = problems.make_ctx(p1, stream=s1), problems.make_ctx(p2, stream=s2)
        for c, p in ((c1, p1), (c2, p2)):
            c.upload(idahip.F_YY, p["yy0"]); c.upload(idahip.F_YP, p["yp0"])
            c.upload(idahip.F_YYPREDICT, p["yy0"]); c.upload(idahip.F_YPPREDICT, p["yp0"])
            c.upload(idahip.F_EWT, np.ones_like(p["yy0"]))
        c2.nls_lsetup(0.0, 100.0)

        def lu():
            c1.nls_lsetup(0.0, 100.0)

        def other(reps=3):
            for _ in range(reps):
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
        print("%-28s LU of %d alone %.2f ms | sys+newton x3 of %d alone %.2f ms | sum %.2f | concurrent %.2f ms" %
              (name, nlu, t_lu, nother, t_ot, t_lu + t_ot, t_both), flush=True)
        c1.close(); c2.close()


if __name__ == "__main__":
    main()
