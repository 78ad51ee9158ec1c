#!/bin/bash
# round 6, session g: the dispatch timeline of the step with one special push per sort cycle and the clutter removed
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$(pwd)/gpurun_out/r6g; mkdir -p $O; ROOTDIR=$(pwd)
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof -o trace -- \
    python $ROOTDIR/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-phase-pass ) > $O/rocprof.log 2>&1
f=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python scripts/kernel_timeline.py $f 200 > $O/timeline_last_steps.txt; tail -2 $O/timeline_last_steps.txt
rm -rf $O/prof
timeout 300 python bench.py --steps 12 --warmup 6 --order 4 --no-cpu-baseline --no-sanity > $O/order4.json 2>/dev/null
python -c "
import json
d=json.load(open('$O/order4.json'))
print('order 4 (global-memory kernels): ms/step %.3f value %.4e' % (d['ms_per_step'], d['value']), {k: round(v['avg_ms'],3) for k,v in d['kernels'].items()})" | tee $O/order4.txt
