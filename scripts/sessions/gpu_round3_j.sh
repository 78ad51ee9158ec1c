#!/bin/bash
# Round-3 evidence in one GPU session: the default bench line, rocprofv3 kernel trace of the same command, PMC traffic
# (FETCH_SIZE / WRITE_SIZE passes) stamped with the kernel sources' fingerprint, then the whole -m gpu suite with its
# slowest tests listed (the driver's round-end step allows it 1200 s) and smoke.
set -u
OUT=$(pwd)/gpurun_out/r3j
mkdir -p $OUT
export TMPDIR=/tmp
ROOTDIR=$(pwd)
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -c 3800 $OUT/bench.json; tail -3 $OUT/bench.err
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- \
    python $ROOTDIR/bench.py --no-cpu-baseline --no-phase-pass --no-sanity ) > $OUT/rocprof.log 2>&1
tail -2 $OUT/rocprof.log
for f in $(find $OUT/prof -name "*kernel_stats*.csv" | head -1); do head -14 $f; cp $f $OUT/kernel_stats.csv; done
timeout 600 python scripts/pmc_traffic.py $OUT/pmc > $OUT/pmc_traffic.log 2>&1
tail -30 $OUT/pmc_traffic.log
rm -rf $OUT/prof/*/*.db $OUT/pmc/*/*/*.db $OUT/prof/*.db $OUT/pmc/*/*.db 2>/dev/null
if [ "${SKIP_TESTS:-0}" != "1" ]; then
timeout ${PYTEST_LIMIT:-1150} python -m pytest tests -m gpu -q -rf --durations=40 2>&1 | tail -70 > $OUT/pytest_gpu.txt
tail -60 $OUT/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
fi
du -sh $OUT
