#!/bin/bash
# Round 4: CKC EvolveB with the row's neighbours through DPP (variants 10-18) against production; bit-exactness on the hardware
set -u
OUT=$(pwd)/gpurun_out/r4z
mkdir -p $OUT
export TMPDIR=/tmp
WXA_PRODUCT_LIB=$(pwd)/warpx_amd/libwarpx_amd_dev.so timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "ckc_bit_exact" 2>&1 | tail -3 | tee $OUT/pytest_ckc_rows.txt
WXA_PRODUCT_LIB=$(pwd)/warpx_amd/libwarpx_amd_dev.so timeout 300 python scripts/ckc_timing.py 256 30 2>&1 | grep -v amdgpu.ids | tee $OUT/ckc_rows_dpp.txt | grep "round 1\|Yee"
