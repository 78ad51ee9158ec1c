#!/bin/bash
# Round 4: deposition phase A with the zero fill behind the offset loads (variant 81) against production.
set -u
OUT=$(pwd)/gpurun_out/r4m
mkdir -p $OUT
export TMPDIR=/tmp
WXA_EXTRA_DEFS=-DWXA_DEV_VARIANTS WXA_LIB_OUT=warpx_amd/libwarpx_amd_dev.so python -m warpx_amd.build --force > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
WXA_PRODUCT_LIB=$(pwd)/warpx_amd/libwarpx_amd_dev.so timeout 400 python scripts/variants.py base WXA_DEPOSIT_VARIANT=81 --repeat 4 2>&1 | grep -v "^\[{" | tail -9 | tee $OUT/deposit_zero_fill_first.txt
WXA_PRODUCT_LIB=$(pwd)/warpx_amd/libwarpx_amd_dev.so timeout 400 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "test_deposit_tile_variants and 81" 2>&1 | tail -3 | tee $OUT/pytest_zf.txt
