#!/bin/bash
# Round 4, second GPU session: the three kernel candidates of scripts/round4/ (prepared from the gfx950 ISA in round 3's
# GPU-less session) in ONE dev build, each against the production configuration, then their parity tests on the MI355X.
# Applies the patches to the working tree of the GPU box's snapshot only (nothing is committed by this script).
#   gpurun --timeout 1200 -- 'bash scripts/sessions/gpu_round4_b.sh'
set -u
OUT=$(pwd)/gpurun_out/r4b
mkdir -p $OUT
export TMPDIR=/tmp
for p in deposit_prefetch_item gather_loads_before_stores stencil_loads_before_stores; do
  patch -p1 --forward -s < scripts/round4/$p.patch || { echo "patch $p does not apply"; exit 1; }
done
WXA_EXTRA_DEFS=-DWXA_DEV_VARIANTS WXA_LIB_OUT=warpx_amd/libwarpx_amd_dev.so python -m warpx_amd.build --force > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
export WXA_PRODUCT_LIB=$(pwd)/warpx_amd/libwarpx_amd_dev.so
# deposition: production against the prefetched work item
timeout 300 python scripts/variants.py base WXA_DEPOSIT_VARIANT=50 --repeat 3 2>&1 | tail -12 | tee $OUT/deposit_prefetch_item.txt
# gather: production (RB 2, PF 3) against PF 4 and PF 5
timeout 300 python scripts/variants.py WXA_GATHER_RB=2,WXA_GATHER_PF=3 WXA_GATHER_RB=2,WXA_GATHER_PF=4 WXA_GATHER_RB=2,WXA_GATHER_PF=5 --repeat 3 2>&1 | tail -16 | tee $OUT/gather_loads_before_stores.txt
# stencils: production (-1), NT on the shared operand (6), the two batched ones (8, 9)
timeout 200 python scripts/stencil_variants.py 256 20 -1,6,8,9 2>&1 | tail -12 | tee $OUT/stencil_loads_before_stores.txt
# parity of the variants on the hardware
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "stencil_configurations" 2>&1 | tail -3 | tee $OUT/pytest_stencil_variants.txt
WXA_DEPOSIT_VARIANT=50 WXA_GATHER_RB=2 WXA_GATHER_PF=5 timeout 400 python -m pytest tests/test_kernels_gpu.py tests/test_step_gpu.py -m gpu -q \
    -k "deposit or esirkepov or gather or test_uniform_plasma_parity" 2>&1 | tail -3 | tee $OUT/pytest_deposit50_gather5.txt
du -sh $OUT
