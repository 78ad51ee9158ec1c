#!/bin/bash
# round 6, session e: merged against separate COUNT / SCATTER pushes, interleaved repeats (the boxes drift by 1-3 %)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$(pwd)/gpurun_out/r6e; mkdir -p $O
for rep in 1 2 3; do for cfg in "1 2" "0 2" "1 3" "0 3"; do set -- $cfg
  WXA_SORT_MERGED=$1 timeout 300 python bench.py --steps 24 --warmup 6 --sort-interval $2 --no-cpu-baseline --no-phase-pass > $O/tmp.json 2>/dev/null
  python -c "
import json
d=json.load(open('$O/tmp.json'))
print('rep $rep merged $1 interval $2: ms/step %.3f' % d['ms_per_step'])"
done; done | tee $O/merged_sort_repeats.txt
rm -f $O/tmp.json
