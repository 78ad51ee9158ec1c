#!/bin/bash
# Round 3, seventh GPU session: deferred particles kept in LDS (phase clocks, timing), the brick tests with the
# one-round particle hand-off (threads of one process on one GPU), the RCCL loop-back, kernel parity of the new scan.
set -u
OUT=$(pwd)/gpurun_out/r3g
mkdir -p $OUT
export TMPDIR=/tmp
ROOTDIR=$(pwd)
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_multibrick_gpu.py -m gpu -q -x 2>&1 | tail -4 > $OUT/pytest.txt
cat $OUT/pytest.txt
timeout 300 python scripts/deposit_profile2.py -1 > $OUT/deposit_phases.txt 2>&1
tail -10 $OUT/deposit_phases.txt
timeout 600 python scripts/variants.py base --repeat 2 > $OUT/base.txt 2> $OUT/base.err
grep -v "^\[" $OUT/base.txt | head -4
