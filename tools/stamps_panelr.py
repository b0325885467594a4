#!/usr/bin/env python3
"""Development tool (timing build -DIDAHIP_STAMPS): phase times of lu_panelr_kernel's first launch (N = 4096 heat Jacobians)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rust-ida_amd"))
import numpy as np
import idahip
from idahip import problems

def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 85
    p = problems.heat1d(n=4096, batch=B)
    ctx = problems.make_ctx(p)
    ctx.upload(idahip.F_YY, p["yy0"]); ctx.upload(idahip.F_YP, p["yp0"])
    stamps = len(sys.argv) <= 2  # any second argument: timing only (a product build has no stamps)
    if stamps:
        ctx.H.idahip_debug_stamps.restype = C.c_void_p
        ctx.H.idahip_debug_stamps.argtypes = [C.c_void_p, C.c_size_t]
        d = ctx.H.idahip_debug_stamps(ctx.h, 8 * B)
    ctx.timing(True)
    for r in range(2):
        ctx.timing_reset()
        rc, info = ctx.nls_lsetup(0.0, 1000.0)
        t = ctx.timing_get()
        print("rep %d: lu %.3f ms (%.1f us/matrix)" % (r, t["lu"]["ms"], t["lu"]["ms"] * 1e3 / B))
    if not stamps:
        return
    raw = ctx.to_host(d, (B, 8), dtype=np.uint64).astype(np.float64) / 100.0
    st = raw[:, :4]
    print("after step 0..3 since loads done (us): %s" % " ".join("%.1f" % (raw[:, 4 + k] - raw[:, 1]).mean() for k in range(4)))
    for i, nm in enumerate(["loads (live, pos, 8 columns)", "8 pivot steps", "epilogue (pos, L11, live list)"]):
        dl = st[:, i + 1] - st[:, i]
        print("%-32s mean %.1f  p10 %.1f p90 %.1f us" % (nm, dl.mean(), *np.percentile(dl, [10, 90])))
    print("kernel span %.1f us" % (st[:, 3].max() - st[:, 0].min()))

if __name__ == "__main__":
    main()
