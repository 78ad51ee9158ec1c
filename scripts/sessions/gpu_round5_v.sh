#!/bin/bash
# Round 5, session v: the streaming deposition with the lane's two particles on one wide frame: tests, config 5's line with
# one sort per step and with two (the second sort makes the frames of a cell's particles start at the same point).
set -u
OUT=$(pwd)/gpurun_out/r5v
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "deposit" 2>&1 | grep -v "^HIP version\|^ROCm version\|^Hostname\|^Librccl\|^RCCL" | tail -2 | tee $OUT/pytest_deposit.txt
show() { python -c "
import json
d=json.load(open('$1'))
print('$2: ms/step %.2f, %.3e particle-steps/s, %.3e cell-updates/s, particles %d -> %d' % (d['ms_per_step'], d['value'], d['cell_updates_per_s'], d['config']['particles_before'], d['config']['particles_after']))
for k,v in d['kernels'].items(): print('  %-18s %.3f ms per launch, %.2f launches per step, %.3f ms per step %s' % (k, v['avg_ms'], v['launches_per_step'], v['ms_per_step'], ('hbm %.3f' % v['hbm_frac']) if 'hbm_frac' in v else ''))
"; }
run() {  # name, env, args
  local name=$1; shift; local envs=$1; shift
  env $envs timeout 400 python scripts/bench_lwfa_boosted.py "$@" > $OUT/$name.json 2> $OUT/$name.err; echo "$name rc=$?"
  show $OUT/$name.json "$name" | tee $OUT/$name.txt
}
run one_sort_per_step "A=1"
run two_sorts_per_step "WXA_SORT_BEHIND_SHIFT=1"
