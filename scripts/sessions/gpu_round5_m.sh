#!/bin/bash
# Round 5, session m: BASELINE config 5 on one GPU with the deposition of a streaming plasma (every particle through the
# wide-frame body inside the tile loop) against the default kernel (WXA_STREAMING_PLASMA=0), at sort intervals 1 and 3.
set -u
OUT=$(pwd)/gpurun_out/r5m
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "deposit or sort_folded" 2>&1 | grep -v "^HIP version\|^ROCm version\|^Hostname\|^Librccl\|^RCCL" | tail -3 | tee $OUT/pytest_deposit.txt
show() { python -c "
import json
d=json.load(open('$1'))
print('$2: ms/step %.2f, %.3e particle-steps/s, %.3e cell-updates/s, particles %d -> %d' % (d['ms_per_step'], d['value'], d['cell_updates_per_s'], d['config']['particles_before'], d['config']['particles_after']))
for k,v in d['kernels'].items(): print('  %-18s %.3f ms per launch, %.2f launches per step, %.3f ms per step %s' % (k, v['avg_ms'], v['launches_per_step'], v['ms_per_step'], ('hbm %.3f' % v['hbm_frac']) if 'hbm_frac' in v else ''))
"; }
run() {  # name, env, args
  local name=$1; shift; local envs=$1; shift
  env $envs timeout 900 python scripts/bench_lwfa_boosted.py "$@" > $OUT/$name.json 2> $OUT/$name.err; echo "$name rc=$?"
  show $OUT/$name.json "$name" | tee $OUT/$name.txt
}
run small_streaming_si1 "A=1" --ncell 64 64 128 --steps 40 --sort-interval 1
run small_default_si1 "WXA_STREAMING_PLASMA=0" --ncell 64 64 128 --steps 40 --sort-interval 1
run big_streaming_si1 "A=1" --sort-interval 1
run big_streaming_si3 "A=1" --sort-interval 3
run big_default_si1 "WXA_STREAMING_PLASMA=0" --sort-interval 1 --steps 12
