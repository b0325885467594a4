#!/usr/bin/env python3
"""Development tool (round 5): the large-n LU pipelines (lu_superpanel_kernel, and round 4's panel-by-panel kernels with
IDAHIP_LU_SUPERPANEL=0) against the oracle on sizes and structures beyond the committed tests' -- dense, mostly zeros, bands that
pivot, a zero pivot late in the matrix, NaN and infinities. Sizes of at most 1024 rows run on the two pipelines of that range
instead (LU variants 4 and 3); `random K` draws K sizes from 9..1024. Prints one line per case; exits non-zero if any case differs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rust-ida_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import idahip
import oracle_lib as O


def gpu_lu(mats, variant=None):
    B, n, _ = mats.shape
    ctx = idahip.Ctx("linear_dense", n, B)
    if variant is not None:
        ctx.set_lu_variant(variant)
    dA = ctx.dev_array(np.ascontiguousarray(np.transpose(mats, (0, 2, 1))))
    dP = ctx.dev_empty(8 * B * n)
    rc, info = ctx.ls_setup(dA, dP, None)
    lu = np.transpose(ctx.to_host(dA, (B, n, n)), (0, 2, 1))
    piv = ctx.to_host(dP, (B, n), dtype=np.int64)
    ctx.close()
    return info, lu, piv


def cases(n, rng):
    i = np.arange(n)
    m = rng.standard_normal((n, n)); yield "dense", m
    m = rng.standard_normal((n, n)); m[np.abs(m) < 1.3] = 0.0; m += np.diag(rng.standard_normal(n) * 0.5); yield "mostly zeros, weak diagonal", m
    m = np.zeros((n, n))
    for d in range(-5, 6):
        k = np.arange(max(0, -d), min(n, n - d)); m[k, k + d] = rng.standard_normal(k.size) * (0.2 if d == 0 else 1.0)
    yield "band of 11 that pivots", m
    m = rng.standard_normal((n, n)); m[:, max(1, n - 70)] = m[:, max(0, n - 71)] * 2.0; yield "dependent columns late (zero pivot or tiny)", m
    m = rng.integers(-2, 3, size=(n, n)).astype(float) + np.eye(n); m[n // 2, min(7, n - 1)] = np.nan; m[3, n // 3] = np.inf; m[n - 5, n // 3] = -np.inf
    yield "integers with ties, NaN, infinities", m


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "random":
        sizes = sorted(int(x) for x in np.random.default_rng(2025).integers(9, 1025, size=int(sys.argv[2])))
    else:
        sizes = [int(a) for a in sys.argv[1:]] or [1029, 1088, 1153, 1283, 2049, 2111, 3071]
    bad = total = 0
    for n in sizes:
        rng = np.random.default_rng(n * 7 + 1)
        mats, names = [], []
        for name, m in cases(n, rng):
            mats.append(m); names.append(name)
        mats = np.array(mats)
        ref = [O.getrf(m) for m in mats]
        for sp in (1, 0):
            if n > 1024:
                os.environ["IDAHIP_LU_SUPERPANEL"] = str(sp)
                info, lu, piv = gpu_lu(mats)
                label = "superpanel " if sp else "panel by panel"
            else:
                info, lu, piv = gpu_lu(mats, 4 if sp else 3)
                label = "wave panel (4)" if sp else "two-row panel (3)"
            for s, name in enumerate(names):
                io, luo, pvo = ref[s]
                ok = info[s] == io and (io != 0 or (np.array_equal(piv[s], pvo) and np.array_equal(lu[s], luo, equal_nan=True)))
                print("n = %4d  %-44s %s  info %d: %s" % (n, name, label, io, "identical" if ok else "DIFFERS"), flush=True)
                bad += 0 if ok else 1
                total += 1
    print("%d of %d cases identical" % (total - bad, total))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
