#!/bin/bash
# Development tool (GPU box): rocprofv3 kernel stats of the batched LU of N = 4096 heat Jacobians (tools/stamps_panelr.py's driver).
# usage: tools/kt_heat.sh <tag> [batch]
TAG=$1; B=${2:-85}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/${ROUND_DIR:-r5}/kth_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT -o kt --output-format csv -- python3 $ROOT/tools/stamps_panelr.py $B nostamps > $OUT/log.txt 2>&1
echo "=== $TAG"; grep "^rep" $OUT/log.txt
python3 - $OUT <<'PY'
import csv, sys, os
for root, _, files in os.walk(sys.argv[1]):
    for f in files:
        if f.endswith("kernel_stats.csv"):
            for r in list(csv.DictReader(open(os.path.join(root, f))))[:8]:
                print("%-90s calls %5s total %9.3f ms avg %8.1f us  %5s%%" % (r["Name"][:90], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
python3 $ROOT/tools/kt_summary.py $OUT 12 | cut -c1-260
find $OUT -name "*.csv" ! -name "*kernel_stats.csv" -delete
