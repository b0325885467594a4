#!/bin/bash
export IDAHIP_ALLOW_TIMING_BUILD=1  # these tools compare builds, timing builds (-DIDAHIP_TIMING_BUILD -DIDAHIP_EXP_...) among them
# Development tool (GPU box): rocprofv3 kernel trace of tools/panel_time.py, per-launch durations of the LU kernels.
# usage: [IDAHIP_LIB_HIP=...] tools/kt.sh <tag> [batch] [round-dir, default r4]
TAG=$1; B=${2:-1370}; RD=${3:-r4}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$RD/kt_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
IDAHIP_GEN_PROCS=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT -o kt --output-format csv -- python3 $ROOT/tools/panel_time.py $B > $OUT/log.txt 2>&1
echo "=== $TAG"; tail -1 $OUT/log.txt
python3 $ROOT/tools/kt_summary.py $OUT 8
find $OUT -name "*.csv" ! -name "*kernel_stats.csv" -delete
