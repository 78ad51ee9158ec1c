#!/bin/bash
# Round 3, second GPU session: gather variants (rows in flight x particle prefetch), LDS atomic lane-grouping microbench,
# the benchmark-regime oracle parity of the new kernels.
set -u
OUT=$(pwd)/gpurun_out/r3b
mkdir -p $OUT
export TMPDIR=/tmp
ROOTDIR=$(pwd)
DEV=$ROOTDIR/warpx_amd/libwarpx_amd_dev.so
WXA_PRODUCT_LIB=$DEV timeout 600 python scripts/variants.py WXA_GATHER_RB=1,WXA_GATHER_PF=0 WXA_GATHER_RB=1,WXA_GATHER_PF=1 WXA_GATHER_RB=1,WXA_GATHER_PF=2 \
    WXA_GATHER_RB=2,WXA_GATHER_PF=1 WXA_GATHER_RB=2,WXA_GATHER_PF=2 WXA_GATHER_RB=0,WXA_GATHER_PF=1 WXA_GATHER_RB=0,WXA_GATHER_PF=2 base --repeat 2 \
    > $OUT/gather_variants.txt 2> $OUT/gather_variants.err
grep -v "^\[" $OUT/gather_variants.txt | head -18; tail -2 $OUT/gather_variants.err
timeout 120 scripts/microbench/lds_atomic_bench > $OUT/lds_atomic_microbench.txt 2>&1
cat $OUT/lds_atomic_microbench.txt
timeout 900 python -m pytest tests/test_step_gpu.py -m gpu -q -x -k "benchmark_regime or uniform_plasma_parity" 2>&1 | tail -4 > $OUT/pytest_step.txt
cat $OUT/pytest_step.txt
du -sh $OUT
