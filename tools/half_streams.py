#!/usr/bin/env python3
"""Development tool (round 5, VERDICT r4 next #1a): config 3 as ONE ensemble of B systems on one stream against TWO ensembles
of B/2 systems on two streams driven by two host threads, the second started half a round later -- does one half's
latency-bound part of a lock-step round (panel kernels' serial chain, Newton passes 2-4, round begin/end) run under the other
half's full launches?   usage: python tools/half_streams.py [B] [rounds]"""
import ctypes as C, os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rust-ida_amd"))
import numpy as np
import idahip
from idahip import problems

hip = C.CDLL("libamdhip64.so")
PROCS = int(os.environ.get("IDAHIP_GEN_PROCS", "16"))


def mkstream():
    s = C.c_void_p()
    assert hip.hipStreamCreateWithFlags(C.byref(s), C.c_uint(1)) == 0  # hipStreamNonBlocking
    return s


FULL = None  # the B systems, generated once; every configuration slices them


class Half:
    def __init__(self, n, batch, first, stagger, stream):
        self.prob = {k: (v[first:first + batch] if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[0] == FULL["yy0"].shape[0] and k not in ("atol", "touts") else v)
                     for k, v in FULL.items()}
        self.ctx = problems.make_ctx(self.prob, stream=stream)
        self.ens = idahip.Ensemble(self.ctx, self.prob["yy0"], self.prob["yp0"])
        self.ens.stream(self.prob["touts"], max(200, 3 * stagger), stagger_rounds=stagger)

    def run(self, k):
        self.ens.stream(self.prob["touts"], k)

    def iters(self):
        return self.ens.total_newton_iters()


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    n, stagger = 512, 62
    if PROCS > 1:
        problems.ensure_fork_server()  # before anything touches the GPU
    global FULL
    FULL = problems.linear_dense(n=n, batch=B, procs=PROCS)
    hip.hipInit(0)
    hip.hipSetDevice(0)
    one = Half(n, B, 0, stagger, mkstream())
    for rep in range(int(os.environ.get("ONE_REPS", "3"))):
        i0 = one.iters(); t0 = time.perf_counter(); one.run(K); dt = time.perf_counter() - t0
        print("one ensemble of %d: %d rounds %.1f ms/round, %.1f k iters/s" % (B, K, dt / K * 1e3, (one.iters() - i0) / dt / 1e3), flush=True)
    one.ens.close(); one.ctx.close(); del one
    # numbers of groups in the order given (a configuration may appear twice: run-to-run spread on one box)
    for cfg in os.environ.get("NGROUPS", "2,3,4").split(","):
        G = int(cfg)
        per = B // G
        if os.environ.get("PLAIN_STREAMS") == "1":
            streams, nconc = [mkstream() for g in range(G)], -1
        else:
            streams, nconc = idahip.concurrent_streams(G)  # probed: really side by side on the device
        share = [[idahip.stream_pair_share(streams[i], streams[j]) for j in range(G)] for i in range(G)]
        print("pairwise share of the chip between the streams (1 = interleaved dispatch): " +
              " | ".join(" ".join("%.2f" % share[i][j] if i != j else "  - " for j in range(G)) for i in range(G)), flush=True)
        groups = [Half(n, per, g * per, stagger, streams[g]) for g in range(G)]
        print("---- %d groups, mutually concurrent streams %d" % (G, nconc), flush=True)
        h = groups[0]
        i0 = h.iters(); t0 = time.perf_counter(); h.run(K); dt = time.perf_counter() - t0
        print("one group of %d alone: %.1f ms/round, %.1f k iters/s" % (per, dt / K * 1e3, (h.iters() - i0) / dt / 1e3), flush=True)
        for delay_ms in (0.0,):
            for rep in range(3):
                i0 = sum(h.iters() for h in groups)
                t0 = time.perf_counter()
                idahip.stream_group([h.ens for h in groups], groups[0].prob["touts"], K, offset_us=int(delay_ms * 1000))  # the product's entry point
                dt = time.perf_counter() - t0
                print("%d groups of %d on %d streams (idaens_stream_group), each %.0f ms after the one before: %.1f ms per round of all, %.1f k iters/s" %
                      (G, per, G, delay_ms, dt / K * 1e3, (sum(h.iters() for h in groups) - i0) / dt / 1e3), flush=True)
        for h in groups:
            h.ens.close(); h.ctx.close()
        del groups, h


if __name__ == "__main__":
    main()
