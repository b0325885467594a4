#!/usr/bin/env python3
"""Development tool: wall time and LU count of every lock-step round of the bench's stream after its spin-up (is the load
stationary, i.e. is a 20-round window representative?). usage: python tools/round_profile.py [rounds] [stagger] [spin_up]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "rust-ida_amd"))
import numpy as np


def main():
    import bench
    from idahip import problems
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 240
    if len(sys.argv) > 2: bench.Runner.STAGGER = int(sys.argv[2])
    if len(sys.argv) > 3: bench.Runner.SPIN_UP = int(sys.argv[3])
    prob = problems.linear_dense(n=512, batch=4096, procs=16)
    r = bench.Runner(prob, 0)
    r.ctx.timing(1)
    ms, lus = [], []
    for i in range(rounds):
        r.ctx.timing_reset()
        r.sync(); t0 = time.perf_counter(); r.step(); r.sync(); ms.append((time.perf_counter() - t0) * 1e3)
        lus.append(r.ctx.timing_get()["lu"]["systems"])
    ms, lus = np.array(ms), np.array(lus)
    for i in range(0, rounds, 10):
        print("rounds %3d-%3d: %.2f ms/round  LU matrices/round %.0f" % (i, i + 9, ms[i:i + 10].mean(), lus[i:i + 10].mean()))
    w = np.array([ms[i:i + 20].mean() for i in range(0, rounds - 19)])
    print("20-round windows: min %.2f  max %.2f  mean %.2f ms/round (spread %.1f %%)" % (w.min(), w.max(), w.mean(), 100 * (w.max() - w.min()) / w.mean()))


if __name__ == "__main__":
    main()
