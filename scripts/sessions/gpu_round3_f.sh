#!/bin/bash
# Round 3, sixth GPU session: the full-size oracle parity tests (256^3) with their printed deviations, and the bench line.
set -u
OUT=$(pwd)/gpurun_out/r3f
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_step_gpu.py -m gpu -q -x -s -k "256" 2>&1 | grep -v "^$" | tail -80 > $OUT/pytest_full_size.txt
cat $OUT/pytest_full_size.txt
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -c 3500 $OUT/bench.json; tail -3 $OUT/bench.err
