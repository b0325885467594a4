#!/bin/bash
export IDAHIP_ALLOW_TIMING_BUILD=1  # these tools compare builds, timing builds (-DIDAHIP_TIMING_BUILD -DIDAHIP_EXP_...) among them
# Development tool (GPU box): A/B of two builds of libidahip on one box. usage: tools/lib_ab.sh <variant> [batch]
# (build B as rust-ida_amd/csrc/libidahip_b.so; the binding takes the library path from IDAHIP_LIB_HIP)
V=${1:-4}; B=${2:-2048}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT
for r in 1 2; do
echo "--- build A"; LU_VARIANT=$V python3 tools/panel_time.py $B 2>&1 | tail -2
echo "--- build B"; IDAHIP_LIB_HIP=$ROOT/rust-ida_amd/csrc/libidahip_b.so LU_VARIANT=$V python3 tools/panel_time.py $B 2>&1 | tail -2
done
