#!/bin/bash
# round 6, session d: one special push per sort cycle (COUNT | SCATTER merged) against a counting and a scattering push, intervals 1 .. 4
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$(pwd)/gpurun_out/r6d; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "sort_folded or bricks_on_one_gpu or overlapped or zero_multi or guard_layer or uniform_plasma_parity or as_processes" 2>&1 | tail -3 | tee $O/pytest.txt
for merged in 1 0; do for si in 1 2 3 4; do
  [ $merged = 0 ] && [ $si = 1 ] && continue
  WXA_SORT_MERGED=$merged timeout 300 python bench.py --steps 12 --warmup 6 --sort-interval $si --no-cpu-baseline > $O/tmp.json 2>/dev/null
  python -c "
import json
d=json.load(open('$O/tmp.json'))
print('merged $merged interval $si: ms/step %.3f' % d['ms_per_step'], {k: round(v['avg_ms'],3) for k,v in d['kernels'].items()}, 'drift %.2e' % d['sanity']['total_energy_drift_over_timed_steps'])"
done; done | tee $O/merged_sort_intervals.txt
rm -f $O/tmp.json
