#!/bin/bash
# Round 4: the deposition's write-back by columns (variant 80) against production; the new CKC production tile in the suite.
#   gpurun --timeout 600 -- 'bash scripts/sessions/gpu_round4_j.sh'
set -u
OUT=$(pwd)/gpurun_out/r4j
mkdir -p $OUT
export TMPDIR=/tmp
WXA_EXTRA_DEFS=-DWXA_DEV_VARIANTS WXA_LIB_OUT=warpx_amd/libwarpx_amd_dev.so python -m warpx_amd.build --force > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
WXA_PRODUCT_LIB=$(pwd)/warpx_amd/libwarpx_amd_dev.so timeout 400 python scripts/variants.py base WXA_DEPOSIT_VARIANT=80 --repeat 4 2>&1 | grep -v "^\[{" | tail -9 | tee $OUT/deposit_flush_by_columns.txt
WXA_PRODUCT_LIB=$(pwd)/warpx_amd/libwarpx_amd_dev.so timeout 400 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "test_deposit_tile_variants and 80" 2>&1 | tail -3 | tee $OUT/pytest_flush.txt
timeout 300 python -m pytest tests/test_kernels_gpu.py tests/test_step_gpu.py -m gpu -q -k "ckc" 2>&1 | tail -3 | tee $OUT/pytest_ckc_production.txt
du -sh $OUT
