#!/bin/bash
# Produces the rocprofv3 evidence committed under profiles/ (run on the GPU box from the repo root):
#   1. kernel trace + stats of `bench.py --steps 40 --warmup 0 --no-cpu-baseline` under IDAHIP_BENCH_TIME_ALL=1 (the HIP-event
#      class timers then cover every launch of the process, as the trace does) and the bench line of the same run;
#   2. the same command under `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes, kernel trace only);
#   3. tools/summarize_profiles.py merges them into one JSON.
# usage: tools/profile_bench.sh <tag>      (outputs under gpurun_out/<tag>/)
set -e
TAG=${1:-prof}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export IDAHIP_GEN_PROCS=1   # no forked generator processes under the profiler
export IDAHIP_BENCH_TIME_ALL=1
ARGS="$ROOT/bench.py --steps 40 --warmup 0 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o kt --output-format csv -- python3 $ARGS > "$OUT/bench.json" 2> "$OUT/kt.err"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/fetch" -o f --output-format csv -- python3 $ARGS > "$OUT/fetch.json" 2> "$OUT/fetch.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/write" -o w --output-format csv -- python3 $ARGS > "$OUT/write.json" 2> "$OUT/write.err"
STATS=$(find "$OUT/kt" -name "*kernel_stats.csv" | head -1)
FETCH=$(find "$OUT/fetch" -name "*counter_collection.csv" | head -1)
WRITE=$(find "$OUT/write" -name "*counter_collection.csv" | head -1)
cp "$STATS" "$OUT/kernel_stats.csv"
python3 "$ROOT/tools/summarize_profiles.py" "$STATS" "$FETCH" "$WRITE" "$OUT/bench.json" "$OUT/summary.json" > "$OUT/summary.txt"
cat "$OUT/summary.txt"
