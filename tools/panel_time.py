#!/usr/bin/env python3
"""Development tool: device time of one batched Jacobian + LU (idahip_nls_lsetup, the copy-free path the stepper uses) on
config-3 matrices. usage: [LU_VARIANT=v] python tools/panel_time.py [batch]; run two variants in one gpurun call for an A/B
(box-to-box variation is a few per cent)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rust-ida_amd"))
import numpy as np
import idahip
from idahip import problems

def main():
    n, B = 512, int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    p = problems.linear_dense(n=n, batch=B, procs=int(os.environ.get("IDAHIP_GEN_PROCS", "16")))
    ctx = problems.make_ctx(p)
    ctx.set_lu_variant(int(os.environ.get("LU_VARIANT", "4")))
    ctx.upload(idahip.F_YY, p["yy0"]); ctx.upload(idahip.F_YP, p["yp0"])
    ctx.timing(True)
    for r in range(3):
        ctx.timing_reset()
        rc, info = ctx.nls_lsetup(0.0, 100.0)
        t = ctx.timing_get()
        print("rep %d: jac %.3f ms  lu %.3f ms  (%.2f us/matrix) info_any=%d" % (r, t["jac"]["ms"], t["lu"]["ms"], t["lu"]["ms"] * 1e3 / B, int(info.any())))


if __name__ == "__main__":  # (the input generator starts worker processes that import this module)
    main()
