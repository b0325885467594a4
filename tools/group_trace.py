#!/usr/bin/env python3
"""Development tool: reads a rocprofv3 kernel trace (csv) of tools/half_streams.py and says, per phase of the run separated by
idle gaps > 50 ms, which hardware queues the kernels ran on, how busy each queue was, and how many kernels were in flight on
average -- i.e. whether the groups of idaens_stream_group really ran side by side.  usage: group_trace.py <kernel_trace.csv>"""
import csv, sys, collections

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"].split("(")[0][-40:]))
rows.sort()
phases, cur = [], [rows[0]]
last_end = rows[0][1]
for r in rows[1:]:
    if r[0] - last_end > 50_000_000:
        phases.append(cur); cur = []
    cur.append(r); last_end = max(last_end, r[1])
phases.append(cur)
for i, ph in enumerate(phases):
    t0, t1 = ph[0][0], max(r[1] for r in ph)
    span = (t1 - t0) / 1e6
    if span < 20 or len(ph) < 500:
        continue
    busy = collections.Counter(); cnt = collections.Counter()
    for s, e, q, name in ph:
        busy[q] += e - s; cnt[q] += 1
    ev = sorted([(s, 1) for s, e, q, n in ph] + [(e, -1) for s, e, q, n in ph])
    depth, lastt, acc, any_busy = 0, ev[0][0], 0.0, 0.0
    for t, d in ev:
        acc += depth * (t - lastt)
        if depth > 0: any_busy += t - lastt
        depth += d; lastt = t
    print("phase %2d: %8.1f ms, %6d kernels, queues %s, mean kernels in flight %.2f, some kernel running %.0f %% of the time" %
          (i, span, len(ph), {q: "%d launches, busy %.0f %%" % (cnt[q], 100.0 * busy[q] / (t1 - t0)) for q in sorted(busy)},
           acc / (t1 - t0), 100.0 * any_busy / (t1 - t0)))
