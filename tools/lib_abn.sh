#!/bin/bash
export IDAHIP_ALLOW_TIMING_BUILD=1  # these tools compare builds, timing builds (-DIDAHIP_TIMING_BUILD -DIDAHIP_EXP_...) among them
# Development tool (GPU box): device time of the batched LU for several builds of libidahip on one box.
# usage: tools/lib_abn.sh <variant> <batch> lib1.so lib2.so ...   (paths relative to rust-ida_amd/csrc)
V=$1; B=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT
for r in 1 2; do
for L in "$@"; do
echo "--- $L"; IDAHIP_LIB_HIP=$ROOT/rust-ida_amd/csrc/$L LU_VARIANT=$V python3 tools/panel_time.py $B 2>&1 | tail -1
done
done
