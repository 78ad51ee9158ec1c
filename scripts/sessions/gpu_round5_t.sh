#!/bin/bash
# Round 5, session t, u: the streaming deposition on 4 x 4 x 5 points (wide along z only; t: crossers in x, y deferred, u: per wave): tests, config 5's line.
set -u
OUT=$(pwd)/gpurun_out/${WXA_SESSION:-r5t}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "deposit" 2>&1 | grep -v "^HIP version\|^ROCm version\|^Hostname\|^Librccl\|^RCCL" | tail -2 | tee $OUT/pytest_deposit.txt
timeout 600 python scripts/bench_lwfa_boosted.py > $OUT/lwfa_boosted.json 2> $OUT/lwfa_boosted.err; echo "line rc=$?"
python -c "
import json
d=json.load(open('$OUT/lwfa_boosted.json'))
print('config 5: ms/step %.2f, %.3e particle-steps/s, %.3e cell-updates/s, particles %d -> %d' % (d['ms_per_step'], d['value'], d['cell_updates_per_s'], d['config']['particles_before'], d['config']['particles_after']))
for k,v in d['kernels'].items(): print('  %-18s %.3f ms per launch, %.2f launches per step, %.3f ms per step %s' % (k, v['avg_ms'], v['launches_per_step'], v['ms_per_step'], ('hbm %.3f' % v['hbm_frac']) if 'hbm_frac' in v else ''))
" | tee $OUT/lwfa_boosted.txt
