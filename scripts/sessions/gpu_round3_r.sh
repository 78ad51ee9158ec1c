#!/bin/bash
# Round 3, seventeenth GPU session: direct deposition on 32-cell chunks (dev variant 50) -- parity, A/B timing.
# (adopted: RowsDirect now uses 32-cell chunks; variant 50 no longer exists)
set -u
OUT=$(pwd)/gpurun_out/r3r
mkdir -p $OUT
export TMPDIR=/tmp
ROOTDIR=$(pwd)
DEV=$ROOTDIR/warpx_amd/libwarpx_amd_dev.so
WXA_DEPOSIT_VARIANT=50 WXA_PRODUCT_LIB=$DEV timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "deposit_current_lds_tiles or fast_and_crossing" 2>&1 | tail -3 > $OUT/pytest_direct_b32.txt
cat $OUT/pytest_direct_b32.txt
WXA_PRODUCT_LIB=$DEV timeout 600 python scripts/variants.py base WXA_DEPOSIT_VARIANT=50 --deposition direct --repeat 3 \
    > $OUT/direct_b32.txt 2> $OUT/direct_b32.err
grep -v "^\[" $OUT/direct_b32.txt | head -8; tail -2 $OUT/direct_b32.err
du -sh $OUT
