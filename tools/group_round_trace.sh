#!/bin/bash
# Development tool (GPU box): rocprofv3 kernel trace of a short `bench.py` run with its groups side by side (NOT the timer mode of
# tools/profile_bench.sh), summarised by tools/group_round_trace.py: the overlap of the groups' launches in the stream.
# usage: tools/group_round_trace.sh <tag> [groups]
TAG=${1:-grt}; G=${2:-4}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/r5/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export IDAHIP_GEN_PROCS=1
timeout -k 10 900 rocprofv3 --kernel-trace -d "$OUT/kt" -o kt --output-format csv -- python3 $ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --groups $G > "$OUT/bench.json" 2> "$OUT/err.txt"
TR=$(find "$OUT/kt" -name "*kernel_trace.csv" | head -1)
python3 $ROOT/tools/group_round_trace.py "$TR" $G > "$OUT/group_round_trace.txt"
rm -rf "$OUT/kt"
head -12 "$OUT/group_round_trace.txt"
