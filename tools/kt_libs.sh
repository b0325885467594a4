#!/bin/bash
export IDAHIP_ALLOW_TIMING_BUILD=1  # these tools compare builds, timing builds (-DIDAHIP_TIMING_BUILD -DIDAHIP_EXP_...) among them
# Development tool (GPU box): tools/kt.sh for several builds of libidahip on one box.
# usage: tools/kt_libs.sh <batch> lib1.so lib2.so ...   (paths relative to rust-ida_amd/csrc)
B=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for L in "$@"; do
IDAHIP_LIB_HIP=$ROOT/rust-ida_amd/csrc/$L $ROOT/tools/kt.sh ${L%.so} $B || exit 1
done
