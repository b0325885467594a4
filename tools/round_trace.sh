#!/bin/bash
# Development tool (GPU box): rocprofv3 kernel trace of a short bench.py stream; prints the launches of the LAST lock-step round in
# order with their durations, so that the cost of a round's small passes (the 2nd..4th Newton iteration and residual launches,
# the control kernels) can be read off. usage: tools/round_trace.sh <tag> [bench.py arguments]
TAG=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/${ROUND_DIR:-r5}/rt_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export IDAHIP_GEN_PROCS=1
timeout -k 10 900 rocprofv3 --kernel-trace -d $OUT -o rt --output-format csv -- python3 $ROOT/bench.py --steps 3 --warmup 0 --no-cpu-baseline --no-extras "$@" > $OUT/bench.json 2> $OUT/err.txt
python3 - "$OUT" <<'P'
import csv, sys, os
p = None
for root, _, files in os.walk(sys.argv[1]):
    for f in files:
        if f.endswith("kernel_trace.csv"):
            p = os.path.join(root, f)
rows = sorted(csv.DictReader(open(p)), key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: n.split("(")[0].replace("void ", "").replace("idahip::", "")[:44]
ends = [i for i, r in enumerate(rows) if "round_end_kernel" in r["Kernel_Name"]]
lo, hi = ends[-4] + 1, ends[-3] + 1   # a round of the first timed repetition
t0 = int(rows[lo]["Start_Timestamp"])
tot = 0.0
for r in rows[lo:hi]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    print("%9.1f us  +%8.1f  %-46s grid %s" % (d, (int(r["Start_Timestamp"]) - t0) / 1e3, short(r["Kernel_Name"]), r.get("Grid_Size", "")))
print("round: %.1f us of kernel time, %.1f us wall" % (tot, (int(rows[hi - 1]["End_Timestamp"]) - t0) / 1e3))
P
find $OUT -name "*.csv" -delete
