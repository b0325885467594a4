#!/usr/bin/env python3
"""Development tool: host/device split of the stream driver (IDAENS_PROFILE=1). usage: IDAENS_PROFILE=1 python tools/host_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rust-ida_amd"))
import numpy as np
import idahip
from idahip import problems


def main():
    p = problems.linear_dense(n=512, batch=4096, procs=16)
    ctx = problems.make_ctx(p)
    ens = idahip.Ensemble(ctx, p["yy0"], p["yp0"])
    ens.stream(p["touts"], 200, stagger_rounds=96)
    t0 = time.perf_counter(); it0 = ens.total_newton_iters()
    ens.stream(p["touts"], 60)
    dt = time.perf_counter() - t0
    print("60 rounds: %.1f ms/round, %.0f iters/s" % (dt / 60 * 1e3, (ens.total_newton_iters() - it0) / dt))
    ens.close()


if __name__ == "__main__":
    main()
