#!/bin/bash
# Round 5, session o: the streaming deposition's wave sums (lanes of a wave that share a frame sum every value over the
# wave, one lane adds it) on the hardware: the kernel tests, then BASELINE config 5 on one GPU at three thresholds.
set -u
OUT=$(pwd)/gpurun_out/r5o
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "streaming_plasma or crowded" 2>&1 | grep -v "^HIP version\|^ROCm version\|^Hostname\|^Librccl\|^RCCL" | tail -3 | tee $OUT/pytest_kernels.txt
show() { python -c "
import json
d=json.load(open('$1'))
print('$2: ms/step %.2f, %.3e particle-steps/s, %.3e cell-updates/s, particles %d -> %d' % (d['ms_per_step'], d['value'], d['cell_updates_per_s'], d['config']['particles_before'], d['config']['particles_after']))
for k,v in d['kernels'].items(): print('  %-18s %.3f ms per launch, %.2f launches per step, %.3f ms per step %s' % (k, v['avg_ms'], v['launches_per_step'], v['ms_per_step'], ('hbm %.3f' % v['hbm_frac']) if 'hbm_frac' in v else ''))
"; }
run() {  # name, env, args
  local name=$1; shift; local envs=$1; shift
  env $envs timeout 600 python scripts/bench_lwfa_boosted.py "$@" > $OUT/$name.json 2> $OUT/$name.err; echo "$name rc=$?"
  show $OUT/$name.json "$name" | tee $OUT/$name.txt
}
run wave_sum_16 "A=1" --ppc 2
run wave_sum_off "WXA_WAVE_SUM_MIN=65" --ppc 2
run wave_sum_6 "WXA_WAVE_SUM_MIN=6" --ppc 2
