#!/usr/bin/env python3
"""Development tool (timing build -DIDAHIP_STAMPS=<k0>): phase times inside lu_trail64w_kernel's launch for super-panel k0
(N = 512 config-3 matrices). usage: IDAHIP_LIB_HIP=.../libidahip_stamps.so python tools/stamps_trail.py [batch] [ncb]"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rust-ida_amd"))
import numpy as np
import idahip
from idahip import problems

def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1312
    ncb = int(sys.argv[2]) if len(sys.argv) > 2 else 7
    p = problems.linear_dense(n=512, batch=B, procs=int(os.environ.get("IDAHIP_GEN_PROCS", "16")))
    ctx = problems.make_ctx(p)
    ctx.upload(idahip.F_YY, p["yy0"]); ctx.upload(idahip.F_YP, p["yp0"])
    nwg = ((B + 7) // 8) * 8 * ncb
    ctx.H.idahip_debug_stamps.restype = C.c_void_p
    ctx.H.idahip_debug_stamps.argtypes = [C.c_void_p, C.c_size_t]
    d = ctx.H.idahip_debug_stamps(ctx.h, 8 * nwg)
    for r in range(2):
        rc, info = ctx.nls_lsetup(0.0, 100.0)
    raw = ctx.to_host(d, (nwg, 8), dtype=np.uint64).astype(np.float64) / 100.0
    raw = raw[raw[:, 0] > 0]
    names = ["start -> L11/live staged + barrier", "solve", "U12 to LDS + barriers", "U12 store issued", "first strip's loads waited for", "first k-chunk (32 pivots)", "rest of the strips"]
    for i, nm in enumerate(names):
        dl = raw[:, i + 1] - raw[:, i]
        print("%-40s mean %6.2f  p10 %6.2f p50 %6.2f p90 %6.2f us" % (nm, dl.mean(), *np.percentile(dl, [10, 50, 90])))
    life = raw[:, 7] - raw[:, 0]
    print("workgroup life mean %.1f p10 %.1f p90 %.1f us; launch span %.1f us; %d workgroups" % (life.mean(), *np.percentile(life, [10, 90]), raw[:, 7].max() - raw[:, 0].min(), len(raw)))
    t0 = raw[:, 0] - raw[:, 0].min()
    order = np.argsort(t0)
    print("start times (us) of workgroups #0, 767, 768, 1535, 1536, last:", " ".join("%.0f" % t0[order[i]] for i in (0, 767, 768, 1535, 1536, len(order) - 1) if i < len(order)))

if __name__ == "__main__":
    main()
