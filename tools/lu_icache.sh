#!/bin/bash
# Development tool (GPU box): instruction-cache counters of the LU kernels of one batched LU (is the 128 KB lu_trail64w_kernel
# held by the 64 KB instruction cache two CUs share?). usage: tools/lu_icache.sh <tag> [batch]
set -e
TAG=${1:-a}; B=${2:-1312}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/r4/icache_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $OUT/avail.txt 2>&1 || true
grep -i -o "SQC\?_[A-Z_]*\(ICACHE\|IFETCH\|INST_CACHE\)[A-Z_]*" $OUT/avail.txt | sort -u > $OUT/icache_counters.txt || true
cat $OUT/icache_counters.txt
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY"; do
  n=$(echo $set | cut -d' ' -f1)
  IDAHIP_GEN_PROCS=1 LU_VARIANT=4 rocprofv3 --kernel-trace --pmc $set -d $OUT/$n -o p --output-format csv -- python3 $ROOT/tools/panel_time.py $B > $OUT/$n.log 2>&1 || { tail -5 $OUT/$n.log; continue; }
  python3 - <<PY
import csv, collections, glob
fs = glob.glob("$OUT/$n/**/*counter_collection.csv", recursive=True)
if fs:
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(fs[0])):
        acc[r["Kernel_Name"][:44]][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, d in acc.items():
        if "lu_" in k:
            print(k)
            for c, v in sorted(d.items()): print("   %-30s %.4g" % (c, v))
PY
done
