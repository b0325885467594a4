import os, sys
sys.path.insert(0, "/root/repo/rust-ida_amd")
import numpy as np, idahip
from idahip import problems
n, B = int(sys.argv[1]), int(sys.argv[2])
p = problems.linear_dense(n=n, batch=B, procs=1)
for sp in (0, 1, 0, 1):
    os.environ["IDAHIP_LU_SUPERPANEL"] = str(sp)
    ctx = problems.make_ctx(p)
    ctx.upload(idahip.F_YY, p["yy0"]); ctx.upload(idahip.F_YP, p["yp0"])
    ctx.timing(True)
    best = 1e9
    for r in range(3):
        ctx.timing_reset(); rc, info = ctx.nls_lsetup(0.0, 100.0); t = ctx.timing_get(); best = min(best, t["lu"]["ms"])
    print("n = %d, %d dense matrices, superpanel %d: LU %.3f ms (%.1f us/matrix)" % (n, B, sp, best, best * 1e3 / B), flush=True)
    ctx.close()
