#!/bin/bash
# Round 4: the deposition's chunk loop without its LDS atomics (111), without its arithmetic (112), without both (113: loads,
# coordinates, frames, deferrals, the phases around the loop) on the production configuration; the wrap kernel with a tile per pass.
set -u
OUT=$(pwd)/gpurun_out/r4u
mkdir -p $OUT
export TMPDIR=/tmp
WXA_PRODUCT_LIB=$(pwd)/warpx_amd/libwarpx_amd_dev.so timeout 500 python scripts/variants.py base WXA_DEPOSIT_VARIANT=111 WXA_DEPOSIT_VARIANT=112 WXA_DEPOSIT_VARIANT=113 --repeat 2 2>&1 | grep -v "^\[{" | tail -9 | tee $OUT/deposit_skeleton.txt
