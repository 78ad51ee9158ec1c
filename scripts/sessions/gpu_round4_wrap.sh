#!/bin/bash
# Round 4: the periodic wrap kernel with four workgroups per face tile -- its line in the kernel trace of a short bench run
set -u
OUT=$(pwd)/gpurun_out/r4wrap
mkdir -p $OUT
export TMPDIR=/tmp
ROOTDIR=$(pwd)
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- \
    python $ROOTDIR/bench.py --steps 12 --no-cpu-baseline --no-phase-pass --no-sanity ) > $OUT/rocprof.log 2>&1
for f in $(find $OUT/prof -name "*kernel_stats*.csv" | head -1); do grep "enforce_periodic\|sort_count\|gather_push_tile" $f | cut -c1-60,200-400; cp $f $OUT/kernel_stats.csv; done
rm -rf $OUT/prof
