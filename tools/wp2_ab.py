#!/usr/bin/env python3
"""Development tool: the batched Jacobian + LU of config-3 matrices with the super-panels factored by one wave per matrix
(IDAHIP_WP2_MIN_SLOTS=0) against two waves per matrix from a given slot count on (lu_wavepanel2.hpp), on one box and on the
same matrices: device time of the LU and of its panel kernels, and the factors bit for bit.
usage: python tools/wp2_ab.py [batch ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rust-ida_amd"))
import numpy as np
import idahip
from idahip import problems


def run(p, B, setting):
    os.environ["IDAHIP_WP2_MIN_SLOTS"] = str(setting)
    sub = {k: (v[:B] if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[0] >= B and k not in ("atol", "touts") else v) for k, v in p.items()}
    ctx = problems.make_ctx(sub)
    ctx.upload(idahip.F_YY, sub["yy0"]); ctx.upload(idahip.F_YP, sub["yp0"])
    best = None
    for level in (1, 2):
        ctx.timing(level)
        for r in range(3):
            ctx.timing_reset()
            rc, info = ctx.nls_lsetup(0.0, 100.0)
            t = ctx.timing_get()
            assert not info.any()
            if level == 1:
                best = t["lu"]["ms"] if best is None else min(best, t["lu"]["ms"])
            else:
                panel, trail = t["lu_panel"]["ms"], t["lu_trail"]["ms"]
    fac = [ctx.download_lu(s) for s in (0, B // 3, B - 1)]
    ctx.close()
    return best, panel, trail, fac


def main():
    batches = [int(a) for a in sys.argv[1:]] or [330, 1312]
    p = problems.linear_dense(n=512, batch=max(batches), procs=int(os.environ.get("IDAHIP_GEN_PROCS", "16")))
    for B in batches:
        ref = None
        for setting in (0, 2, 3, 4, 5, 7, 0):
            lu, panel, trail, fac = run(p, B, setting)
            same = True
            if ref is None:
                ref = fac
            else:
                same = all(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) for a, b in zip(fac, ref))
            print("B = %4d  two waves from %d slots on: LU %.3f ms (%.2f us/matrix)  panel kernels %.3f ms  trailing %.3f ms  factors %s" %
                  (B, setting, lu, lu * 1e3 / B, panel, trail, "identical" if same else "DIFFER"), flush=True)


if __name__ == "__main__":
    main()
