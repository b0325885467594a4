#!/usr/bin/env python3
"""Micro-benchmark of the batched LU (idahip_ls_setup) and solve on random matrices: kernel-class device time by HIP events.
usage: python tools/lu_bench.py [n] [batch] [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rust-ida_amd"))
import numpy as np
import idahip

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
rng = np.random.default_rng(0)
mats = rng.standard_normal((B, n, n))
ctx = idahip.Ctx("linear_dense", n, B)
dP = ctx.dev_empty(8 * B * n)
dA = ctx.dev_array(mats)
ctx.timing(True)
for r in range(reps):
    ctx._chk(ctx.H.idahip_memcpy_h2d(ctx.h, dA, mats.ctypes.data, mats.nbytes), "h2d")
    ctx.timing_reset()
    t0 = time.perf_counter()
    rc, info = ctx.ls_setup(dA, dP)
    dt = time.perf_counter() - t0
    ms = ctx.timing_get()["lu"]["ms"]
    flops = B * (2.0 / 3.0) * n ** 3
    print("rep %d: LU n=%d B=%d  device %.3f ms (wall %.3f ms)  %.2f us/matrix  %.2f TFLOP/s  alg %.1f GB/s  info_any=%d"
          % (r, n, B, ms, dt * 1e3, ms * 1e3 / B, flops / (ms * 1e-3) / 1e12, B * (16 * n * n + 8 * n) / (ms * 1e-3) / 1e9, int(info.any())))
