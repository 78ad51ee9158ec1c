#!/bin/bash
# Round 4, third GPU session: the deposition with hole filling (an empty slot of the direct part takes a late pair of a
# cell of the same bank class) and persistent tiles (one workgroup per CU loops over tiles), each alone and together,
# against round 3's kernel (variant 60), in one dev build; then their parity tests on the MI355X.
#   gpurun --timeout 900 -- 'bash scripts/sessions/gpu_round4_c.sh'
set -u
OUT=$(pwd)/gpurun_out/r4c
mkdir -p $OUT
export TMPDIR=/tmp
WXA_EXTRA_DEFS=-DWXA_DEV_VARIANTS WXA_LIB_OUT=warpx_amd/libwarpx_amd_dev.so python -m warpx_amd.build --force > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
export WXA_PRODUCT_LIB=$(pwd)/warpx_amd/libwarpx_amd_dev.so
timeout 400 python scripts/variants.py base WXA_DEPOSIT_VARIANT=60 WXA_DEPOSIT_VARIANT=64 WXA_DEPOSIT_VARIANT=63 WXA_DEPOSIT_VARIANT=61 --repeat 3 2>&1 | grep -v "^\[{" | tail -16 | tee $OUT/deposit_hole_filling_persistent_tiles.txt
timeout 400 python -m pytest tests/test_kernels_gpu.py tests/test_step_gpu.py -m gpu -q \
    -k "deposit or esirkepov or test_uniform_plasma_parity" 2>&1 | tail -3 | tee $OUT/pytest_deposit.txt
du -sh $OUT
