#!/bin/bash
# Round 3, twelfth GPU session: the chunks of the deposition handed out through an LDS counter (dev variant 40) -- parity,
# A/B timing against the static shares.
set -u
OUT=$(pwd)/gpurun_out/r3l
mkdir -p $OUT
export TMPDIR=/tmp
ROOTDIR=$(pwd)
DEV=$ROOTDIR/warpx_amd/libwarpx_amd_dev.so
WXA_PRODUCT_LIB=$DEV timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "tile_variants and 40" 2>&1 | tail -4 > $OUT/pytest_variants.txt
cat $OUT/pytest_variants.txt
WXA_PRODUCT_LIB=$DEV timeout 600 python scripts/variants.py base WXA_DEPOSIT_VARIANT=40 --repeat 4 \
    > $OUT/deposit_dyn.txt 2> $OUT/deposit_dyn.err
grep -v "^\[" $OUT/deposit_dyn.txt | head -10; tail -2 $OUT/deposit_dyn.err
du -sh $OUT
