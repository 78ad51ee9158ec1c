#!/bin/bash
# Round 5, session e: the recorded keys one free-flight step ahead (WXA_SORT_PREDICT=0: of the positions themselves), two
# interleaved repeats at sort intervals 3 and 2; the whole -m gpu suite; BASELINE config 5 on one GPU as a throughput line.
set -u
OUT=$(pwd)/gpurun_out/r5e
mkdir -p $OUT
export TMPDIR=/tmp
line() { python -c "
import json,sys
d=json.load(open('$1'))
print('$2', 'ms/step %.3f value %.4e' % (d['ms_per_step'], d['value']), {k: round(v['avg_ms'],3) for k,v in d['kernels'].items()})"; }
for rep in 1 2; do
  for si in 3 2; do
    WXA_SORT_PREDICT=0 timeout 300 python bench.py --steps 24 --warmup 6 --sort-interval $si --no-cpu-baseline --no-sanity > $OUT/nopredict_si${si}_$rep.json 2> $OUT/nopredict_si${si}_$rep.err
    line $OUT/nopredict_si${si}_$rep.json "keys of the positions si$si rep $rep"
    timeout 300 python bench.py --steps 24 --warmup 6 --sort-interval $si --no-cpu-baseline --no-sanity > $OUT/predict_si${si}_$rep.json 2> $OUT/predict_si${si}_$rep.err
    line $OUT/predict_si${si}_$rep.json "keys one free-flight step ahead si$si rep $rep"
  done
done 2>&1 | tee $OUT/ab.txt
timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -v "^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -8 | tee $OUT/pytest_gpu.txt
timeout 900 python scripts/bench_lwfa_boosted.py > $OUT/lwfa_boosted_bench.json 2> $OUT/lwfa_boosted_bench.err
python -c "
import json
d=json.load(open('$OUT/lwfa_boosted_bench.json'))
print('config 5 on one GPU: ms/step %.2f, %.3e particle-steps/s, %.3e cell-updates/s, particles %d -> %d' % (d['ms_per_step'], d['value'], d['cell_updates_per_s'], d['config']['particles_before'], d['config']['particles_after']))
for k,v in d['kernels'].items(): print('  %-18s %.3f ms per launch, %.2f launches per step, %.3f ms per step %s' % (k, v['avg_ms'], v['launches_per_step'], v['ms_per_step'], ('hbm %.3f' % v['hbm_frac']) if 'hbm_frac' in v else ''))
" | tee $OUT/lwfa_boosted_bench.txt
tail -3 $OUT/lwfa_boosted_bench.err
