#!/bin/bash
# Round 3, eighth GPU session: the whole -m gpu suite except the three full-size cases (run in session r3f), then smoke.
set -u
OUT=$(pwd)/gpurun_out/r3h
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x -k "not 256" 2>&1 | tail -12 > $OUT/pytest_gpu.txt
cat $OUT/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
