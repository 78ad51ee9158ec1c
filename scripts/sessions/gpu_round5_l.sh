#!/bin/bash
# Round 5, session l: a crowded cell's excess pairs as chunks of the tile loop (sessions j, k: 600-690 ms per deposition at 256 x 256 x 512): the deposition tests,
# the boosted wakefield deck at the size that faulted (64 x 64 x 128 x 8 per cell) and at 256 x 256 x 512 (BASELINE config
# 5 on one GPU), and the headline line twice (the check must not cost anything).
set -u
OUT=$(pwd)/gpurun_out/r5l
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "deposit or sort_folded" 2>&1 | grep -v "^HIP version\|^ROCm version\|^Hostname\|^Librccl\|^RCCL" | tail -3 | tee $OUT/pytest_deposit.txt
show() { python -c "
import json
d=json.load(open('$1'))
print('$2: ms/step %.2f, %.3e particle-steps/s, %.3e cell-updates/s, particles %d -> %d' % (d['ms_per_step'], d['value'], d['cell_updates_per_s'], d['config']['particles_before'], d['config']['particles_after']))
for k,v in d['kernels'].items(): print('  %-18s %.3f ms per launch, %.2f launches per step, %.3f ms per step %s' % (k, v['avg_ms'], v['launches_per_step'], v['ms_per_step'], ('hbm %.3f' % v['hbm_frac']) if 'hbm_frac' in v else ''))
"; }
timeout 300 python scripts/bench_lwfa_boosted.py --ncell 64 64 128 --steps 40 > $OUT/lwfa_small.json 2> $OUT/lwfa_small.err; echo "small rc=$?"
show $OUT/lwfa_small.json "64 x 64 x 128" | tee $OUT/lwfa_small.txt
timeout 900 python scripts/bench_lwfa_boosted.py > $OUT/lwfa_boosted_bench.json 2> $OUT/lwfa_boosted_bench.err; echo "big rc=$?"
show $OUT/lwfa_boosted_bench.json "256 x 256 x 512" | tee $OUT/lwfa_boosted_bench.txt
tail -2 $OUT/lwfa_boosted_bench.err | cut -c1-200
for rep in 1 2; do
  timeout 300 python bench.py --steps 24 --warmup 6 --no-cpu-baseline --no-sanity > $OUT/bench_$rep.json 2> $OUT/bench_$rep.err
  python -c "
import json
d=json.load(open('$OUT/bench_$rep.json'))
print('headline rep $rep', 'ms/step %.3f value %.4e' % (d['ms_per_step'], d['value']), {k: round(v['avg_ms'],3) for k,v in d['kernels'].items()})"
done | tee $OUT/headline.txt
