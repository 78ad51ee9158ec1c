#!/bin/bash
# Round 3, eleventh GPU session: 10 and 11 waves per deposition tile (dev variants 30, 31) -- parity, A/B timing; wave 0's
# share of the chunk loop (profile build); the one test session r3j found failing.
set -u
OUT=$(pwd)/gpurun_out/r3k
mkdir -p $OUT
export TMPDIR=/tmp
ROOTDIR=$(pwd)
DEV=$ROOTDIR/warpx_amd/libwarpx_amd_dev.so
timeout 300 python -m pytest tests/test_step_gpu.py -m gpu -q -x -k "particle_boundaries" 2>&1 | tail -3 > $OUT/pytest_fixed.txt
cat $OUT/pytest_fixed.txt
WXA_PRODUCT_LIB=$DEV timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "tile_variants" 2>&1 | tail -4 > $OUT/pytest_variants.txt
cat $OUT/pytest_variants.txt
WXA_PRODUCT_LIB=$DEV timeout 600 python scripts/variants.py base WXA_DEPOSIT_VARIANT=30 WXA_DEPOSIT_VARIANT=31 --repeat 3 \
    > $OUT/deposit_waves.txt 2> $OUT/deposit_waves.err
grep -v "^\[" $OUT/deposit_waves.txt | head -10; tail -2 $OUT/deposit_waves.err
timeout 300 python scripts/deposit_profile2.py -1 > $OUT/deposit_phases.txt 2>&1
tail -11 $OUT/deposit_phases.txt
du -sh $OUT
