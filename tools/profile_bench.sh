#!/bin/bash
# Produces the rocprofv3 evidence committed under profiles/ (run on the GPU box from the repo root):
#   1. kernel trace + stats of `bench.py --steps 40 --warmup 0 --no-cpu-baseline` under IDAHIP_BENCH_TIME_ALL=1 (the HIP-event
#      class timers then cover every launch of the process, as the trace does) and the bench line of the same run;
#   2. the same command under `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes, kernel trace only);
#   3. the same command under two SQ-counter passes (issue / wait / LDS counters per kernel);
#   4. tools/summarize_profiles.py merges them into one JSON (with the hash of the kernel sources: bench.py only quotes the
#      traffic of a summary taken with the kernels it runs).
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
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_SCA -d "$OUT/sq1" -o s --output-format csv -- python3 $ARGS > "$OUT/sq1.json" 2> "$OUT/sq1.err"
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d "$OUT/sq2" -o s --output-format csv -- python3 $ARGS > "$OUT/sq2.json" 2> "$OUT/sq2.err"
SQ1=$(find "$OUT/sq1" -name "*counter_collection.csv" | head -1)
SQ2=$(find "$OUT/sq2" -name "*counter_collection.csv" | head -1)
STATS=$(find "$OUT/kt" -name "*kernel_stats.csv" | head -1)
FETCH=$(find "$OUT/fetch" -name "*counter_collection.csv" | head -1)
WRITE=$(find "$OUT/write" -name "*counter_collection.csv" | head -1)
cp "$STATS" "$OUT/kernel_stats.csv"
python3 "$ROOT/tools/summarize_profiles.py" "$STATS" "$FETCH" "$WRITE" "$OUT/bench.json" "$OUT/summary.json" $SQ1 $SQ2 > "$OUT/summary.txt"
# the raw per-launch counter files are large: only the summaries travel back
rm -rf "$OUT/kt" "$OUT/fetch" "$OUT/write" "$OUT/sq1" "$OUT/sq2"
cat "$OUT/summary.txt"
