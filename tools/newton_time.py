import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "rust-ida_amd"))
import numpy as np, idahip
from idahip import problems

def main():
    n, B = 512, int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    p = problems.linear_dense(n=n, batch=B, procs=int(os.environ.get("IDAHIP_GEN_PROCS", "16")))
    ctx = problems.make_ctx(p)
    ctx.upload(idahip.F_YY, p["yy0"]); ctx.upload(idahip.F_YP, p["yp0"])
    ctx.upload(idahip.F_EWT, np.ones_like(p["yy0"]))
    ctx.nls_lsetup(0.0, 100.0)
    ctx.upload(idahip.F_DELTA, np.random.default_rng(0).standard_normal(p["yy0"].shape))
    ctx.timing(True)
    for r in range(4):
        ctx.timing_reset()
        ctx.newton_iter(1.0)
        t = ctx.timing_get()["newton_iter"]
        print("rep %d: newton_iter %.3f ms (%.1f GB/s)" % (r, t["ms"], (8*n*n+40*n) * B / t["ms"] / 1e6))


if __name__ == "__main__":  # (the input generator starts worker processes that import this module)
    main()
