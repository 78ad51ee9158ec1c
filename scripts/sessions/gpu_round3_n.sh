#!/bin/bash
# Round 3, fourteenth GPU session: the gather's 64-particle chunks handed out through an LDS counter (WXA_GATHER_PF=3,
# dev build) -- parity, A/B timing against the static shares (PF=2), alone and with the deposition's variant 40.
set -u
OUT=$(pwd)/gpurun_out/r3n
mkdir -p $OUT
export TMPDIR=/tmp
ROOTDIR=$(pwd)
DEV=$ROOTDIR/warpx_amd/libwarpx_amd_dev.so
WXA_GATHER_RB=2 WXA_GATHER_PF=3 WXA_PRODUCT_LIB=$DEV timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "gather_push" 2>&1 | tail -3 > $OUT/pytest_gather_dyn.txt
cat $OUT/pytest_gather_dyn.txt
WXA_PRODUCT_LIB=$DEV timeout 600 python scripts/variants.py WXA_GATHER_RB=2,WXA_GATHER_PF=2 WXA_GATHER_RB=2,WXA_GATHER_PF=3 WXA_GATHER_RB=2,WXA_GATHER_PF=3,WXA_DEPOSIT_VARIANT=40 --repeat 4 \
    > $OUT/gather_dyn.txt 2> $OUT/gather_dyn.err
grep -v "^\[" $OUT/gather_dyn.txt | head -12; tail -2 $OUT/gather_dyn.err
du -sh $OUT
