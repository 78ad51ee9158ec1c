#!/bin/bash
# Round 3, sixteenth GPU session: PushPX + DepositCurrent in one tile kernel (wxa_push_and_deposit) -- parity on the GPU,
# A/B timing against the two kernels on the bench regime's particle state (one and two steps after a sort).
set -u
OUT=$(pwd)/gpurun_out/r3p
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "push_and_deposit or deposit_current_lds or gather_push_lds" 2>&1 | tail -6 > $OUT/pytest_fused.txt
cat $OUT/pytest_fused.txt
for S in 1 2; do
  timeout 300 python scripts/fused_timing.py 256 $S 3 2>&1 | grep -v amdgpu.ids | tail -4 | tee -a $OUT/fused_timing.txt
done
du -sh $OUT
