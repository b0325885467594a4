#!/usr/bin/env python3
"""Development tool: per-launch durations (us) of the LU kernels from a rocprofv3 kernel trace csv. usage: kt_summary.py <dir or csv> [last N]"""
import csv, sys, os, collections
p = sys.argv[1]
if os.path.isdir(p):
    for root, _, files in os.walk(p):
        for f in files:
            if f.endswith("kernel_trace.csv"):
                p = os.path.join(root, f)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
d = collections.OrderedDict()
for r in csv.DictReader(open(p)):
    d.setdefault(r["Kernel_Name"][:70], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in d.items():
    if any(s in k for s in ("trail", "wavepanel", "finalize", "panel", "u12", "update16")):
        print("%-72s n=%4d  last: %s  sum(last)=%.0f" % (k, len(v), " ".join("%.0f" % x for x in v[-n:]), sum(v[-n:])))
