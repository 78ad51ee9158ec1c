#!/bin/bash
# Round 4: even particle orders on the tiles (each particle on its own frame; the lone list and the direct deposition's pair
# frames are for odd orders): the deposition tests on the hardware, the secondary configurations as bench lines
set -u
OUT=$(pwd)/gpurun_out/r4even
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_step_gpu.py -m gpu -q -rf -k "deposit or esirkepov or direct_vay_ckc or fp32" 2>&1 | grep -E "passed|failed|FAILED|Error" | tee $OUT/pytest.txt
SESSION=r4even bash scripts/sessions/gpu_round4_l.sh
