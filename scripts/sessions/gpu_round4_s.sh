#!/bin/bash
# Round 4: the production deposition with 16 waves per workgroup at 128 VGPRs (20 spilled dwords in the chunk loop)
set -u
OUT=$(pwd)/gpurun_out/r4s
mkdir -p $OUT
export TMPDIR=/tmp
WXA_PRODUCT_LIB=$(pwd)/warpx_amd/libwarpx_amd_dev.so timeout 500 python scripts/variants.py base WXA_DEPOSIT_VARIANT=90 --repeat 3 2>&1 | grep -v "^\[{" | tail -8 | tee $OUT/deposit_16_waves.txt
WXA_PRODUCT_LIB=$(pwd)/warpx_amd/libwarpx_amd_dev.so timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "variants and 90" 2>&1 | tail -2 | tee $OUT/pytest_variant90.txt
