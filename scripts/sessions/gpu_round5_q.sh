#!/bin/bash
# Round 5, session q: a tile whose tail table overflows keeps no tail (all pairs beyond the fourth are excess chunks) and
# the excess chunks take their pairs from sixteen streams: the deposition tests, config 5 with the kernels of its last steps,
# the headline twice.
set -u
OUT=$(pwd)/gpurun_out/r5q
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "deposit" 2>&1 | grep -v "^HIP version\|^ROCm version\|^Hostname\|^Librccl\|^RCCL" | tail -3 | tee $OUT/pytest_kernels.txt
cd /tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python $GRAFT_REPO_ROOT/scripts/bench_lwfa_boosted.py --steps 30 > $OUT/lwfa_boosted.json 2> $OUT/lwfa_boosted.err; echo "trace rc=$?"
cd $GRAFT_REPO_ROOT
python -c "
import json
d=json.load(open('$OUT/lwfa_boosted.json'))
print('config 5 (under rocprofv3): ms/step %.2f, %.3e particle-steps/s, %.3e cell-updates/s, particles %d -> %d' % (d['ms_per_step'], d['value'], d['cell_updates_per_s'], d['config']['particles_before'], d['config']['particles_after']))
for k,v in d['kernels'].items(): print('  %-18s %.3f ms per launch, %.2f launches per step, %.3f ms per step %s' % (k, v['avg_ms'], v['launches_per_step'], v['ms_per_step'], ('hbm %.3f' % v['hbm_frac']) if 'hbm_frac' in v else ''))
" | tee $OUT/lwfa_boosted.txt
f=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python - <<PY | tee $OUT/kernels_of_the_last_steps.txt
import csv, collections
rows = []
with open("$f") as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:70]))
rows.sort()
rows = rows[-600:]
acc = collections.defaultdict(lambda: [0, 0.0])
for s, e, k in rows:
    acc[k][0] += 1; acc[k][1] += (e - s) / 1e6
for k, (c, ms) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:14]:
    print("%8.2f ms %5d calls %8.3f ms each  %s" % (ms, c, ms / c, k))
PY
rm -rf $OUT/trace
for rep in 1 2; do
  timeout 300 python bench.py --steps 24 --warmup 6 --no-cpu-baseline --no-sanity > $OUT/bench_$rep.json 2> $OUT/bench_$rep.err
  python -c "
import json
d=json.load(open('$OUT/bench_$rep.json'))
print('headline rep $rep', 'ms/step %.3f value %.4e' % (d['ms_per_step'], d['value']), {k: round(v['avg_ms'],3) for k,v in d['kernels'].items()})"
done | tee $OUT/headline.txt
