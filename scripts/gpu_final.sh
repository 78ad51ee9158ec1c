#!/bin/bash
# Round-end evidence in one GPU-box session: full parity suite, smoke, the default bench line and a
# rocprofv3 kernel trace of the same bench command (copy the results into profiles/).
set -u
mkdir -p gpurun_out/final
export TMPDIR=/tmp
ROOTDIR=$(pwd)
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -4 > gpurun_out/final/pytest_gpu.txt
cat gpurun_out/final/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err
tail -c 600 gpurun_out/final/bench.json
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOTDIR/gpurun_out/final/prof -o trace -- \
    python $ROOTDIR/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-phase-pass ) > gpurun_out/final/rocprof.log 2>&1
tail -2 gpurun_out/final/rocprof.log
for f in $(find gpurun_out/final/prof -name "*kernel_stats*.csv" | head -1); do head -16 $f; done
