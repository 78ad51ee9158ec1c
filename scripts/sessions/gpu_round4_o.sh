#!/bin/bash
# Round 4: the gather's stragglers collected per tile and written as one block (WXA_GATHER_SL=1) against production; the
# deposition's new production configuration (write-back by columns, zero fill first, lone partners on their fast frame) through
# the kernel tests and the full-size parity tests.
set -u
OUT=$(pwd)/gpurun_out/r4o
mkdir -p $OUT
export TMPDIR=/tmp
WXA_EXTRA_DEFS=-DWXA_DEV_VARIANTS WXA_LIB_OUT=warpx_amd/libwarpx_amd_dev.so python -m warpx_amd.build --force > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
WXA_PRODUCT_LIB=$(pwd)/warpx_amd/libwarpx_amd_dev.so timeout 500 python scripts/variants.py WXA_GATHER_RB=2,WXA_GATHER_PF=3 WXA_GATHER_RB=2,WXA_GATHER_PF=3,WXA_GATHER_SL=1 --repeat 4 2>&1 | grep -v "^\[{" | tail -9 | tee $OUT/gather_stragglers_by_tile.txt
WXA_PRODUCT_LIB=$(pwd)/warpx_amd/libwarpx_amd_dev.so WXA_GATHER_RB=2 WXA_GATHER_PF=3 WXA_GATHER_SL=1 timeout 400 python -m pytest tests/test_kernels_gpu.py tests/test_step_gpu.py -m gpu -q -k "gather or test_uniform_plasma_parity" 2>&1 | tail -3 | tee $OUT/pytest_gather_sl.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_step_gpu.py -m gpu -q -k "deposit or esirkepov or 256_against or benchmark_regime" 2>&1 | tail -3 | tee $OUT/pytest_deposit_production.txt
