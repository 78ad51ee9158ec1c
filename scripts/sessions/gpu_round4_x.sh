#!/bin/bash
# Round 4: the deposition's tail table ordered by cell and its chunks interleaved with the direct chunks (93)
set -u
OUT=$(pwd)/gpurun_out/r4x
mkdir -p $OUT
export TMPDIR=/tmp
WXA_PRODUCT_LIB=$(pwd)/warpx_amd/libwarpx_amd_dev.so timeout 500 python scripts/variants.py base WXA_DEPOSIT_VARIANT=94 --repeat 3 2>&1 | grep -v "^\[{" | tail -7 | tee $OUT/deposit_tile_blocks.txt
WXA_PRODUCT_LIB=$(pwd)/warpx_amd/libwarpx_amd_dev.so timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "variants and 94" 2>&1 | tail -2 | tee $OUT/pytest_variants.txt
