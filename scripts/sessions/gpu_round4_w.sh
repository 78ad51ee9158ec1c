#!/bin/bash
# Round 4: what the deposition's skeleton (3.8 of 5.9 ms) consists of: 115 = the phases around the chunk loop alone,
# 114 = + the loop's work items and particle loads, 113 = + coordinates, frames, deferrals (no pair body)
set -u
OUT=$(pwd)/gpurun_out/r4w
mkdir -p $OUT
export TMPDIR=/tmp
WXA_PRODUCT_LIB=$(pwd)/warpx_amd/libwarpx_amd_dev.so timeout 500 python scripts/variants.py base WXA_DEPOSIT_VARIANT=115 WXA_DEPOSIT_VARIANT=114 WXA_DEPOSIT_VARIANT=113 --repeat 2 2>&1 | grep -v "^\[{" | tail -9 | tee $OUT/deposit_skeleton_parts.txt
