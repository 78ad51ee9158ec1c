#!/bin/bash
# Round 4: CKC EvolveB with the E planes requested two steps ahead (variants 6-9) against production and the older tile shapes.
#   gpurun --timeout 600 -- 'bash scripts/sessions/gpu_round4_i.sh'
set -u
OUT=$(pwd)/gpurun_out/r4i
mkdir -p $OUT
export TMPDIR=/tmp
WXA_EXTRA_DEFS=-DWXA_DEV_VARIANTS WXA_LIB_OUT=warpx_amd/libwarpx_amd_dev.so python -m warpx_amd.build --force > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
export WXA_PRODUCT_LIB=$(pwd)/warpx_amd/libwarpx_amd_dev.so
timeout 300 python scripts/ckc_timing.py 256 30 2>&1 | grep -v amdgpu.ids | tee $OUT/ckc_planes_two_steps_ahead.txt
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "ckc" 2>&1 | tail -3 | tee $OUT/pytest_ckc.txt
du -sh $OUT
