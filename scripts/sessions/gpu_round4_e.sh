#!/bin/bash
# Round 4, fifth GPU session: two more candidates, each against production in one dev build --
#  * deposition: the direct chunks' cell offsets from global memory instead of LDS (variants 65, 66: + dynamic chunks);
#  * gather: lanes l and l ^ 1 exchange through DPP and store 16 bytes each (WXA_GATHER_ST=1): 3 dwordx4 stores instead of 6 dwordx2.
#   gpurun --timeout 900 -- 'bash scripts/sessions/gpu_round4_e.sh'
set -u
OUT=$(pwd)/gpurun_out/r4e
mkdir -p $OUT
export TMPDIR=/tmp
WXA_EXTRA_DEFS=-DWXA_DEV_VARIANTS WXA_LIB_OUT=warpx_amd/libwarpx_amd_dev.so python -m warpx_amd.build --force > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
export WXA_PRODUCT_LIB=$(pwd)/warpx_amd/libwarpx_amd_dev.so
timeout 400 python scripts/variants.py base WXA_DEPOSIT_VARIANT=65 WXA_DEPOSIT_VARIANT=66 WXA_GATHER_RB=2,WXA_GATHER_PF=3 WXA_GATHER_RB=2,WXA_GATHER_PF=3,WXA_GATHER_ST=1 --repeat 3 2>&1 | grep -v "^\[{" | tail -16 | tee $OUT/deposit_global_offsets_gather_paired_stores.txt
WXA_GATHER_RB=2 WXA_GATHER_PF=3 WXA_GATHER_ST=1 timeout 400 python -m pytest tests/test_kernels_gpu.py tests/test_step_gpu.py -m gpu -q \
    -k "test_deposit_tile_variants or gather or test_uniform_plasma_parity" 2>&1 | tail -3 | tee $OUT/pytest_variants.txt
du -sh $OUT
