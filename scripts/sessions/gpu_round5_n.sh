#!/bin/bash
# Round 5, session n: tiles shared by several workgroups (heavy_tiles.hpp) on the hardware: the kernel tests that split
# tiles (WXA_HEAVY_TILE), BASELINE config 5 on one GPU with and without the splitting, the headline twice.
set -u
OUT=$(pwd)/gpurun_out/r5n
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "deposit or sort_folded or gather_push" 2>&1 | grep -v "^HIP version\|^ROCm version\|^Hostname\|^Librccl\|^RCCL" | tail -3 | tee $OUT/pytest_kernels.txt
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
run big_shared_tiles "A=1"
run big_shared_tiles_8k "WXA_HEAVY_TILE=8192"
for rep in 1 2; do
  timeout 300 python bench.py --steps 24 --warmup 6 --no-cpu-baseline --no-sanity > $OUT/bench_$rep.json 2> $OUT/bench_$rep.err
  python -c "
import json
d=json.load(open('$OUT/bench_$rep.json'))
print('headline rep $rep', 'ms/step %.3f value %.4e' % (d['ms_per_step'], d['value']), {k: round(v['avg_ms'],3) for k,v in d['kernels'].items()})"
  WXA_HEAVY_TILE=0 timeout 300 python bench.py --steps 24 --warmup 6 --no-cpu-baseline --no-sanity > $OUT/bench_off_$rep.json 2> $OUT/bench_off_$rep.err
  python -c "
import json
d=json.load(open('$OUT/bench_off_$rep.json'))
print('headline, splitting off rep $rep', 'ms/step %.3f value %.4e' % (d['ms_per_step'], d['value']), {k: round(v['avg_ms'],3) for k,v in d['kernels'].items()})"
done | tee $OUT/headline.txt
