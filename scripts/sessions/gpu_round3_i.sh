#!/bin/bash
# Round 3, ninth GPU session: why the oracle leg of the 128^3 benchmark-regime test takes minutes on the box.
set -u
OUT=$(pwd)/gpurun_out/r3i
mkdir -p $OUT
export TMPDIR=/tmp
{
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null; lscpu | grep -E "^CPU\(s\)|Thread|Socket|Model name"
echo OMP_NUM_THREADS=${OMP_NUM_THREADS:-unset}
timeout 200 python scripts/oracle_time.py 64 3 random 2>&1 | tail -1
timeout 300 python scripts/oracle_time.py 128 3 random 2>&1 | tail -1
OMP_NUM_THREADS=32 timeout 300 python scripts/oracle_time.py 128 3 random 2>&1 | tail -1
OMP_NUM_THREADS=16 timeout 300 python scripts/oracle_time.py 128 3 random 2>&1 | tail -1
} > $OUT/oracle_time.txt 2>&1
cat $OUT/oracle_time.txt
timeout 600 python -m pytest tests/test_step_gpu.py -m gpu -q -x -k "benchmark_regime" --durations=3 2>&1 | tail -8 > $OUT/pytest_regime.txt
cat $OUT/pytest_regime.txt
