#!/bin/bash
# round 6, session zi: the sort interval of the headline once more, in turns on one box (2 is bench.py's default; 3 came out ahead in
# three of the four evidence sessions' single runs)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$(pwd)/gpurun_out/r6zi; mkdir -p $O
for rep in 1 2 3; do for k in 2 3 4; do
  timeout 300 python bench.py --steps 24 --warmup 12 --no-cpu-baseline --no-sanity --sort-interval $k > $O/tmp.json 2>/dev/null
  python -c "
import json
d=json.load(open('$O/tmp.json'))
print('rep $rep interval $k:', 'ms/step %.3f' % d['ms_per_step'], {k: round(v['avg_ms'],3) for k,v in d['kernels'].items() if k in ('GatherAndPush','CurrentDeposition')})"
done; done | tee $O/sort_interval_repeats.txt
rm -f $O/tmp.json
