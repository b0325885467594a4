#!/usr/bin/env python3
"""Development tool (round 5): why does the rate of four groups side by side differ by ~6 % between boxes whose stand-alone kernel
times agree to 1 %? Runs one ensemble of 4096 config-3 systems and four groups of 1024 for K rounds each while sampling rocm-smi
(shader clock, socket power) every 50 ms, and prints rate, mean clock and mean power of each phase plus the box's GPU identity."""
import os, subprocess, sys, threading, time, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rust-ida_amd"))
import numpy as np
import idahip
from idahip import problems


class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.stop = False
        self.rows = []

    def run(self):
        while not self.stop:
            try:
                out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--csv"], capture_output=True, text=True, timeout=5).stdout
                sclk = re.search(r"\((\d+)Mhz\)", out)
                nums = re.findall(r"(\d+\.\d+)", out)
                self.rows.append((time.perf_counter(), out.strip().replace("\n", " | ")[:300]))
            except Exception as e:
                self.rows.append((time.perf_counter(), "error %s" % e))
            time.sleep(0.05)


def main():
    B, K, n, stagger = 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 60, 512, 62
    procs = int(os.environ.get("IDAHIP_GEN_PROCS", "16"))
    if procs > 1:
        problems.ensure_fork_server()
    full = problems.linear_dense(n=n, batch=B, procs=procs)
    print(subprocess.run(["rocm-smi", "--showproductname", "--showserial", "--showperflevel", "--showmaxpower"], capture_output=True, text=True).stdout[-1500:], flush=True)

    def build(G):
        per = B // G
        streams, nconc = idahip.concurrent_streams(G) if G > 1 else ([None], 1)
        subs = [{k: (v[g * per:(g + 1) * per] if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[0] == B and k not in ("atol", "touts") else v) for k, v in full.items()} for g in range(G)]
        ctxs = [problems.make_ctx(s, stream=streams[g]) for g, s in enumerate(subs)]
        enss = [idahip.Ensemble(c, s["yy0"], s["yp0"]) for c, s in zip(ctxs, subs)]
        idahip.stream_group(enss, full["touts"], 200, stagger_rounds=stagger) if G > 1 else enss[0].stream(full["touts"], 200, stagger_rounds=stagger)
        return ctxs, enss
    for G in (1, 4, 1, 4):
        ctxs, enss = build(G)
        smp = Sampler(); smp.start()
        time.sleep(0.3)
        i0 = sum(e.total_newton_iters() for e in enss); t0 = time.perf_counter()
        if G > 1:
            idahip.stream_group(enss, full["touts"], K)
        else:
            enss[0].stream(full["touts"], K)
        dt = time.perf_counter() - t0
        smp.stop = True; smp.join()
        rows = [r for t, r in smp.rows if t >= t0 and t <= t0 + dt]
        print("G = %d: %.1f k iters/s over %.2f s; %d rocm-smi samples; first / middle / last:" % (G, (sum(e.total_newton_iters() for e in enss) - i0) / dt / 1e3, dt, len(rows)), flush=True)
        for r in (rows[:1] + rows[len(rows) // 2:len(rows) // 2 + 1] + rows[-1:]):
            print("    " + r, flush=True)
        for e in enss: e.close()
        for c in ctxs: c.close()


if __name__ == "__main__":
    main()
