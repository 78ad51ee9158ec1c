#!/bin/bash
# Round-3 closing evidence at HEAD: the default bench line, sort-interval sweep, rocprofv3 kernel trace of the bench
# command, PMC traffic stamped with the kernel sources' fingerprint, the whole -m gpu suite, smoke.
set -u
OUT=$(pwd)/gpurun_out/r3o
mkdir -p $OUT
export TMPDIR=/tmp
ROOTDIR=$(pwd)
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -c 1500 $OUT/bench.json | head -c 300; echo; python -c "import json;d=json.load(open('$OUT/bench.json'));print(d['ms_per_step'],d['value'],{k:round(v['avg_ms'],3) for k,v in d['kernels'].items()})"; tail -3 $OUT/bench.err
for S in 2 4 5; do
  timeout 300 python bench.py --sort-interval $S --no-cpu-baseline --no-phase-pass > $OUT/bench_sort$S.json 2>> $OUT/bench.err
  python -c "import json;d=json.load(open('$OUT/bench_sort$S.json'));print('sort interval $S:',d['ms_per_step'],d['value'])"
done
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- \
    python $ROOTDIR/bench.py --no-cpu-baseline --no-phase-pass --no-sanity ) > $OUT/rocprof.log 2>&1
tail -2 $OUT/rocprof.log
for f in $(find $OUT/prof -name "*kernel_stats*.csv" | head -1); do head -8 $f | cut -c1-160; cp $f $OUT/kernel_stats.csv; done
timeout 600 python scripts/pmc_traffic.py $OUT/pmc > $OUT/pmc_traffic.log 2>&1
tail -24 $OUT/pmc_traffic.log
rm -rf $OUT/prof/*/*.db $OUT/pmc/*/*/*.db $OUT/prof/*.db $OUT/pmc/*/*.db $OUT/prof/*kernel_trace.csv 2>/dev/null
timeout ${PYTEST_LIMIT:-1150} python -m pytest tests -m gpu -q -rf --durations=8 2>&1 | tail -30 > $OUT/pytest_gpu.txt
tail -22 $OUT/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
du -sh $OUT
