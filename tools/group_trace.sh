#!/bin/bash
# Development tool (GPU box): rocprofv3 kernel trace of tools/half_streams.py for the configurations in NGROUPS (default 4,2,4),
# summarised per phase by tools/group_trace.py. usage: tools/group_trace.sh <tag> [B] [rounds]
TAG=${1:-gt}; B=${2:-4096}; K=${3:-20}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/r5/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export IDAHIP_GEN_PROCS=1 ONE_REPS=1
export NGROUPS=${NGROUPS:-4,2,4}
rocprofv3 --kernel-trace -d "$OUT/kt" -o kt --output-format csv -- python3 $ROOT/tools/half_streams.py $B $K > "$OUT/run.log" 2> "$OUT/run.err"
TR=$(find "$OUT/kt" -name "*kernel_trace.csv" | head -1)
python3 $ROOT/tools/group_trace.py "$TR" > "$OUT/phases.txt"
head -2 "$TR" > "$OUT/trace_head.csv"
rm -rf "$OUT/kt"
cat "$OUT/run.log" | grep -v "^one ens"; cat "$OUT/phases.txt"
