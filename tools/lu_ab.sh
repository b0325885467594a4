#!/bin/bash
# Development tool (GPU box): LU parity tests, then device time of the batched LU for the listed variants, then a
# rocprofv3 kernel trace of the last one. usage: tools/lu_ab.sh "3 4" [batch]
set -e
VARIANTS=${1:-"3 4"}
B=${2:-2048}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $ROOT/gpurun_out/ab
cd $ROOT
for v in $VARIANTS; do
  LU_VARIANT=$v python3 tools/panel_time.py $B 2>&1 | tee gpurun_out/ab/time_v$v.txt
done
last=${VARIANTS##* }
cd /tmp && export TMPDIR=/tmp
IDAHIP_GEN_PROCS=1 LU_VARIANT=$last rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/ab/kt -o kt --output-format csv -- python3 $ROOT/tools/panel_time.py $B > $ROOT/gpurun_out/ab/kt.log 2>&1
cp $(find $ROOT/gpurun_out/ab/kt -name "*kernel_stats.csv" | head -1) $ROOT/gpurun_out/ab/kernel_stats_v$last.csv
head -12 $ROOT/gpurun_out/ab/kernel_stats_v$last.csv
