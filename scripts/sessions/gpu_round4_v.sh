#!/bin/bash
# Round 4: the deposition's next chunk requested in the middle of the current chunk's pair body (91: before the last
# component, 92: before the second)
set -u
OUT=$(pwd)/gpurun_out/r4v
mkdir -p $OUT
export TMPDIR=/tmp
WXA_PRODUCT_LIB=$(pwd)/warpx_amd/libwarpx_amd_dev.so timeout 500 python scripts/variants.py base WXA_DEPOSIT_VARIANT=91 WXA_DEPOSIT_VARIANT=92 --repeat 3 2>&1 | grep -v "^\[{" | tail -10 | tee $OUT/deposit_next_chunk_requested_mid_body.txt
WXA_PRODUCT_LIB=$(pwd)/warpx_amd/libwarpx_amd_dev.so timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "variants and (91 or 92)" 2>&1 | tail -2 | tee $OUT/pytest_variants.txt
