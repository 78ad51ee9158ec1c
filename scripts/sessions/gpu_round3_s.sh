#!/bin/bash
# Round 3, eighteenth GPU session: CKC's EvolveB with single ds_read_b64 into a register window (dev variants 6, 7) --
# (the variant was slower and its code was removed again: profiles/round3/README.md; this script is the record of how r3s_* was produced)
# bit-exactness on the GPU, timing against the accessor version (variant -1 = production, 5).
set -u
OUT=$(pwd)/gpurun_out/r3s
mkdir -p $OUT
export TMPDIR=/tmp
ROOTDIR=$(pwd)
DEV=$ROOTDIR/warpx_amd/libwarpx_amd_dev.so
for V in 6 7; do
WXA_CKC_VARIANT=$V WXA_PRODUCT_LIB=$DEV timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_step_gpu.py -m gpu -q -x -k "ckc" 2>&1 | tail -2 | tee -a $OUT/pytest_ckc_asm.txt
done
WXA_PRODUCT_LIB=$DEV timeout 300 python scripts/ckc_timing.py 2>&1 | grep -v amdgpu.ids | tee $OUT/ckc_timing.txt
du -sh $OUT
