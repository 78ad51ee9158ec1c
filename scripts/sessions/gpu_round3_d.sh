#!/bin/bash
# Round 3, fourth GPU session: where the gather tile kernel's time goes -- the kernel without its staging, without its
# stores, and the staging alone (timing experiments of the dev build; they leave the particles unpushed).
set -u
OUT=$(pwd)/gpurun_out/r3d
mkdir -p $OUT
export TMPDIR=/tmp
ROOTDIR=$(pwd)
DEV=$ROOTDIR/warpx_amd/libwarpx_amd_dev.so
WXA_PRODUCT_LIB=$DEV timeout 600 python scripts/variants.py WXA_GATHER_RB=1,WXA_GATHER_PF=0 WXA_GATHER_RB=1,WXA_GATHER_PF=9 WXA_GATHER_RB=1,WXA_GATHER_PF=7 \
    WXA_GATHER_RB=1,WXA_GATHER_PF=8 WXA_GATHER_RB=1,WXA_GATHER_PF=0 --repeat 1 > $OUT/gather_parts.txt 2> $OUT/gather_parts.err
grep -v "^\[" $OUT/gather_parts.txt | head -8; tail -2 $OUT/gather_parts.err
