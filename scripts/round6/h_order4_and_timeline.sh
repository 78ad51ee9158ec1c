#!/bin/bash
# round 6, session h: order 4 on the LDS tiles (tests + bench line), the step's dispatch timeline at HEAD
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$(pwd)/gpurun_out/r6h; mkdir -p $O; ROOTDIR=$(pwd)
timeout 900 python -m pytest tests -m gpu -x -q -k "gather_push_lds_tiles or deposit_current_lds_tiles or test_uniform_plasma_parity or esirkepov_continuity" 2>&1 | tail -3 | tee $O/pytest_order4.txt
for o in 4 2 1; do
timeout 300 python bench.py --steps 12 --warmup 6 --order $o --no-cpu-baseline --no-sanity > $O/tmp.json 2>/dev/null
python -c "
import json
d=json.load(open('$O/tmp.json'))
print('order $o: ms/step %.3f value %.4e' % (d['ms_per_step'], d['value']), {k: round(v['avg_ms'],3) for k,v in d['kernels'].items()})"
done | tee $O/orders.txt
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof -o trace -- \
    python $ROOTDIR/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-phase-pass --no-sanity ) > $O/rocprof.log 2>&1
f=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python scripts/kernel_timeline.py $f 120 > $O/timeline_last_steps.txt; tail -2 $O/timeline_last_steps.txt
rm -rf $O/prof $O/tmp.json
