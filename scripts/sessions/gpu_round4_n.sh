#!/bin/bash
# Round 4: deposition phase D with the lone partners on their fast frame (82), and with the zero fill first as well (83).
set -u
OUT=$(pwd)/gpurun_out/r4n
mkdir -p $OUT
export TMPDIR=/tmp
WXA_EXTRA_DEFS=-DWXA_DEV_VARIANTS WXA_LIB_OUT=warpx_amd/libwarpx_amd_dev.so python -m warpx_amd.build --force > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
WXA_PRODUCT_LIB=$(pwd)/warpx_amd/libwarpx_amd_dev.so timeout 500 python scripts/variants.py base WXA_DEPOSIT_VARIANT=82 WXA_DEPOSIT_VARIANT=83 WXA_DEPOSIT_VARIANT=81 --repeat 4 2>&1 | grep -v "^\[{" | tail -17 | tee $OUT/deposit_lone_partners_fast_frame.txt
WXA_PRODUCT_LIB=$(pwd)/warpx_amd/libwarpx_amd_dev.so timeout 400 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "test_deposit_tile_variants and (82 or 83)" 2>&1 | tail -3 | tee $OUT/pytest_singles.txt
