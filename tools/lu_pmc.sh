#!/bin/bash
# Development tool (GPU box): SQ counters of the LU kernels of one batched LU. usage: tools/lu_pmc.sh <variant> [batch]
set -e
V=${1:-4}; B=${2:-2048}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/pmc; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
IDAHIP_GEN_PROCS=1 LU_VARIANT=$V rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU -d $OUT/p1 -o p --output-format csv -- python3 $ROOT/tools/panel_time.py $B > $OUT/p1.log 2>&1
python3 - <<PY
import csv, collections, glob
f = glob.glob("$OUT/p1/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:44]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()): print("   %-22s %.4g" % (c, v))
PY
