#!/bin/bash
# One GPU-box session: parity tests, smoke, a short bench and a rocprofv3 kernel trace.
# Usage (from the repo root on the GPU box): bash scripts/gpu_check.sh [steps] [warmup]
set -u
STEPS=${1:-3}
WARM=${2:-1}
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest -m gpu" | tee gpurun_out/summary.txt
timeout 900 python -m pytest tests -m gpu -x -q ${PYTEST_EXTRA:-} 2>&1 | tail -40 | tee gpurun_out/pytest_gpu.log
echo "pytest exit: ${PIPESTATUS[0]}" | tee -a gpurun_out/summary.txt
echo "== smoke" | tee -a gpurun_out/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee gpurun_out/smoke.log
echo "== bench" | tee -a gpurun_out/summary.txt
timeout 900 python bench.py --steps $STEPS --warmup $WARM 2>gpurun_out/bench.err | tee gpurun_out/bench.json
tail -5 gpurun_out/bench.err
echo "== rocprofv3 kernel trace" | tee -a gpurun_out/summary.txt
ROOTDIR=$(pwd)
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOTDIR/gpurun_out/prof -o trace -- \
    python $ROOTDIR/bench.py --steps $STEPS --warmup $WARM --no-cpu-baseline --no-phase-pass ) > gpurun_out/rocprof.log 2>&1
tail -3 gpurun_out/rocprof.log
find gpurun_out/prof -name "*kernel_stats*" | head
for f in $(find gpurun_out/prof -name "*kernel_stats*.csv" | head -1); do head -25 $f; done
