#!/bin/bash
# Round 4: the lane's two particles as one 16-byte load per array (95), and the loop's loads alone with them (116 vs 114)
set -u
OUT=$(pwd)/gpurun_out/r4y
mkdir -p $OUT
export TMPDIR=/tmp
WXA_PRODUCT_LIB=$(pwd)/warpx_amd/libwarpx_amd_dev.so timeout 500 python scripts/variants.py base WXA_DEPOSIT_VARIANT=95 WXA_DEPOSIT_VARIANT=114 WXA_DEPOSIT_VARIANT=116 --repeat 3 2>&1 | grep -v "^\[{" | tail -13 | tee $OUT/deposit_16_byte_loads.txt
WXA_PRODUCT_LIB=$(pwd)/warpx_amd/libwarpx_amd_dev.so timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "variants and 95" 2>&1 | tail -2 | tee $OUT/pytest_variants.txt
