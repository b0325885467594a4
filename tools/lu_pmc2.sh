#!/bin/bash
# Development tool (GPU box): LDS / memory-side counters of the LU kernels. usage: tools/lu_pmc2.sh <variant> [batch]
V=${1:-4}; B=${2:-2048}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/pmc2; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { # name, counters...
  n=$1; shift
  IDAHIP_GEN_PROCS=1 LU_VARIANT=$V rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$n -o p --output-format csv -- python3 $ROOT/tools/panel_time.py $B > $OUT/$n.log 2>&1
}
run a SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM
run b SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAVES SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_BUSY_CU_CYCLES
run c GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum
python3 - <<PY
import csv, collections, glob
for n in "abc":
    fs = glob.glob("$OUT/%s/**/*counter_collection.csv" % n, recursive=True)
    if not fs: print(n, "no output"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(fs[0])):
        acc[r["Kernel_Name"][:40]][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, d in acc.items():
        if "lu_" not in k: continue
        print(k)
        for c, v in sorted(d.items()): print("   %-26s %.4g" % (c, v))
PY
