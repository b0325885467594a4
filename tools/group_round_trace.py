#!/usr/bin/env python3
"""Development tool: from a rocprofv3 kernel trace (csv) of `bench.py --groups G`, the stretch in which all G group streams run
side by side (the stream's warm-up and timed rounds): which hardware queue ran what, how many kernels were in flight, how
much of the time a kernel of the batched LU ran beside an HBM-bound kernel of another group, and an excerpt of the timeline
(one column per queue).  usage: group_round_trace.py <kernel_trace.csv> [G] [excerpt_ms]"""
import csv, sys, collections

path = sys.argv[1]
G = int(sys.argv[2]) if len(sys.argv) > 2 else 4
EXC = float(sys.argv[3]) if len(sys.argv) > 3 else 14.0
short = lambda n: n.split("(")[0].replace("void ", "").replace("idahip::", "")
rows = []
for r in csv.DictReader(open(path)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), short(r["Kernel_Name"])))
rows.sort()
byq = collections.defaultdict(list)
for s, e, q, n in rows:
    byq[q].append((s, e, n))
# the G busiest queues over the whole trace are the candidates; a millisecond is "all active" when each has a launch within +-3 ms
t_lo, t_hi = rows[0][0], max(r[1] for r in rows)
import bisect
starts = {q: [s for s, e, n in v] for q, v in byq.items()}
def active(q, t):
    i = bisect.bisect_left(starts[q], t - 3_000_000)
    return i < len(starts[q]) and starts[q][i] <= t + 3_000_000
best, cur = (0, 0), None
t = t_lo
while t < t_hi:
    na = sum(1 for q in byq if active(q, t))
    if na >= G:
        cur = (cur[0], t) if cur else (t, t)
        if cur[1] - cur[0] > best[1] - best[0]:
            best = cur
    else:
        cur = None
    t += 1_000_000
w0, w1 = best
print("all %d group streams active for %.0f ms of the trace (%.0f ms in all)" % (G, (w1 - w0) / 1e6, (t_hi - t_lo) / 1e6))
w0 += 10_000_000; w1 -= 5_000_000  # away from the edges
win = [r for r in rows if r[0] >= w0 and r[1] <= w1]
span = w1 - w0
LU = ("lu_",)
HBM = ("linear_sys_kernel", "newton_iter_kernel")
cls = lambda n: "lu" if n.startswith(LU) else ("hbm" if n.startswith(HBM) else "other")
ev = []
for s, e, q, n in win:
    ev.append((s, 1, cls(n), q)); ev.append((e, -1, cls(n), q))
ev.sort()
depth = collections.Counter(); last = w0
acc_depth = 0.0; t_any = 0.0; t_two = 0.0; t_lu_hbm = 0.0; t_lu_lu = 0.0; t_hbm_hbm = 0.0
for tt, d, c, q in ev:
    dt = tt - last
    tot = sum(depth.values())
    acc_depth += tot * dt
    if tot > 0: t_any += dt
    if tot > 1: t_two += dt
    if depth["lu"] > 0 and depth["hbm"] > 0: t_lu_hbm += dt
    if depth["lu"] > 1: t_lu_lu += dt
    if depth["hbm"] > 1: t_hbm_hbm += dt
    depth[c] += d; last = tt
qs = sorted(byq, key=lambda q: -sum(1 for r in win if r[2] == q))[:G]
print("window of %.1f ms: %d launches on queues %s" % (span / 1e6, len(win), {q: sum(1 for r in win if r[2] == q) for q in qs}))
print("mean kernels in flight %.2f; some kernel running %.1f %% of the time, two or more %.1f %%" % (acc_depth / span, 100 * t_any / span, 100 * t_two / span))
print("an LU kernel beside an HBM-bound kernel (residual / Newton iteration) of another group %.1f %% of the time; two LU kernels %.1f %%; two HBM-bound kernels %.1f %%" %
      (100 * t_lu_hbm / span, 100 * t_lu_lu / span, 100 * t_hbm_hbm / span))
busy = collections.Counter()
for s, e, q, n in win:
    busy[(q, cls(n))] += e - s
for q in qs:
    print("  queue %s: LU %.0f %%, HBM-bound %.0f %%, other %.0f %% of the window" % (q, 100 * busy[(q, "lu")] / span, 100 * busy[(q, "hbm")] / span, 100 * busy[(q, "other")] / span))
# excerpt: one column per queue, a line per launch in start order
e0 = w0 + span // 2
print("\nexcerpt, %.0f ms from the middle of the window (start time in us; one column per queue: kernel and duration in us)" % EXC)
col = {q: i for i, q in enumerate(qs)}
for s, e, q, n in win:
    if s < e0 or s > e0 + EXC * 1e6 or q not in col:
        continue
    cell = "%s %.0f" % (n.replace("_kernel", "")[:24], (e - s) / 1e3)
    print("%9.1f  %s%s" % ((s - e0) / 1e3, " " * (34 * col[q]), cell))
