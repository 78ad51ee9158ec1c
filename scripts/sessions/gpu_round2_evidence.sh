#!/bin/bash
# Round-2 evidence in one GPU session: whole parity suite, smoke, the default bench line, rocprofv3 kernel trace of the
# same command, PMC traffic (FETCH_SIZE / WRITE_SIZE passes) stamped with the kernel sources' fingerprint.
set -u
OUT=$(pwd)/gpurun_out/r2ev
mkdir -p $OUT
export TMPDIR=/tmp
ROOTDIR=$(pwd)
timeout 1500 python -m pytest tests -m gpu -q -rf 2>&1 | tail -40 > $OUT/pytest_gpu.txt
tail -6 $OUT/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -c 3000 $OUT/bench.json; tail -3 $OUT/bench.err
if [ "${SKIP_PROF:-0}" != "1" ]; then
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- \
    python $ROOTDIR/bench.py --no-cpu-baseline --no-phase-pass --no-sanity ) > $OUT/rocprof.log 2>&1
tail -2 $OUT/rocprof.log
for f in $(find $OUT/prof -name "*kernel_stats*.csv" | head -1); do head -14 $f; cp $f $OUT/kernel_stats.csv; done
timeout 900 python scripts/pmc_traffic.py $OUT/pmc > $OUT/pmc_traffic.log 2>&1
tail -30 $OUT/pmc_traffic.log
rm -rf $OUT/prof/*/*.db $OUT/pmc/*/*/*.db 2>/dev/null
fi
du -sh $OUT
