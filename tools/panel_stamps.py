#!/usr/bin/env python3
"""Development tool: per-phase cycle stamps of lu_panel_kernel (needs csrc/libidahip_stamps.so)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rust-ida_amd"))
import numpy as np
import idahip
idahip.LIB_HIP = os.path.join(ROOT, "rust-ida_amd", "csrc", "libidahip_stamps.so")
n, B = 512, int(sys.argv[1]) if len(sys.argv) > 1 else 2048
rng = np.random.default_rng(0)
mats = rng.standard_normal((B, n, n))
ctx = idahip.Ctx("linear_dense", n, B)
dA = ctx.dev_array(mats); dP = ctx.dev_empty(8 * B * n)
os.environ["IDAHIP_LU_DBG"] = "31"
ctx.ls_setup(dA, dP)
buf = np.zeros(8 * 8 * 8, dtype=np.uint64)
ctx.H.idahip_debug_ubuf.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
ctx.H.idahip_debug_ubuf(ctx.h, buf.ctypes.data_as(C.c_void_p), buf.nbytes)
st = buf.reshape(8, 8, 8).astype(np.int64)
names = ["loads", "reduce", "publish", "barrier", "scan", "rowstore", "update", "epilogue"]
for wg in (0, 3):
    for wv in (0, 3, 7):
        print("wg %d wave %d: " % (wg, wv) + "  ".join("%s %d" % (nm, v) for nm, v in zip(names, st[wg, wv])), " | per step:", (st[wg, wv, 1:7] // 32).tolist())
