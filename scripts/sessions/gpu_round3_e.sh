#!/bin/bash
# Round 3, fifth GPU session: the gather with plain (cached) prefetch loads.
set -u
OUT=$(pwd)/gpurun_out/r3e
mkdir -p $OUT
export TMPDIR=/tmp
ROOTDIR=$(pwd)
DEV=$ROOTDIR/warpx_amd/libwarpx_amd_dev.so
WXA_PRODUCT_LIB=$DEV timeout 600 python scripts/variants.py WXA_GATHER_RB=1,WXA_GATHER_PF=0 WXA_GATHER_RB=1,WXA_GATHER_PF=2 WXA_GATHER_RB=2,WXA_GATHER_PF=2 \
    WXA_GATHER_RB=3,WXA_GATHER_PF=2 WXA_GATHER_RB=1,WXA_GATHER_PF=1 WXA_GATHER_RB=0,WXA_GATHER_PF=2 --repeat 2 > $OUT/gather_pf.txt 2> $OUT/gather_pf.err
grep -v "^\[" $OUT/gather_pf.txt | head -14; tail -2 $OUT/gather_pf.err
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "gather" 2>&1 | tail -3
