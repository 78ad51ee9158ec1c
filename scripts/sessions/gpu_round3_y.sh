#!/bin/bash
# Round 3, GPU session y: workgroups of the gather's straggler pass (WXA_STRAGGLER_BLOCKS, dev build) -- A/B timing.
# (no effect; the switch was removed again -- this script is the record of how r3y_* was produced)
set -u
OUT=$(pwd)/gpurun_out/r3y
mkdir -p $OUT
export TMPDIR=/tmp
DEV=$(pwd)/warpx_amd/libwarpx_amd_dev.so
WXA_PRODUCT_LIB=$DEV timeout 600 python scripts/variants.py WXA_STRAGGLER_BLOCKS=512 WXA_STRAGGLER_BLOCKS=1280 WXA_STRAGGLER_BLOCKS=2560 WXA_STRAGGLER_BLOCKS=5120 --repeat 3 \
    > $OUT/straggler_blocks.txt 2> $OUT/straggler_blocks.err
grep -v "^\[" $OUT/straggler_blocks.txt | head -12; tail -2 $OUT/straggler_blocks.err
