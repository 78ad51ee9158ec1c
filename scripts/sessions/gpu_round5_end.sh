#!/bin/bash
# Round 5, the evidence session at the round's last kernel sources (after the shared tiles): smoke, the whole -m gpu suite, the PMC passes (traffic
# + SQ, stamped), the default bench line (with the CPU baseline) as the driver runs it, rocprofv3 --kernel-trace --stats of
# the same command, the secondary configurations, BASELINE config 5 on one GPU.
set -u
OUT=$(pwd)/gpurun_out/r5fin2
mkdir -p $OUT
export TMPDIR=/tmp
ROOTDIR=$(pwd)
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $OUT/smoke.txt
timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -v "^HIP version\|^ROCm version\|^Hostname\|^Librccl\|^RCCL" | tail -8 | tee $OUT/pytest_gpu.txt
timeout 1200 python scripts/pmc_traffic.py $OUT/pmc > $OUT/pmc_stdout.txt 2>&1; echo "pmc rc=$?"
cp $OUT/pmc/r5_pmc_counters.json profiles/round5/r5_pmc_counters.json 2>/dev/null && cp $OUT/pmc/r5_pmc_counters.json $OUT/r5_pmc_counters.json
rm -rf $OUT/pmc/FETCH_SIZE $OUT/pmc/WRITE_SIZE $OUT/pmc/sq1 $OUT/pmc/sq2
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
tail -c 1500 $OUT/bench.json
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- \
    python $ROOTDIR/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-phase-pass ) > $OUT/rocprof.log 2>&1
for f in $(find $OUT/prof -name "*kernel_stats*.csv" | head -1); do cp $f $OUT/kernel_stats.csv; head -14 $f; done
rm -rf $OUT/prof
for cfg in "--pusher vay" "--deposition direct" "--order 2" "--order 1"; do
  timeout 300 python bench.py --steps 12 --warmup 6 --no-cpu-baseline --no-sanity $cfg > $OUT/tmp.json 2>/dev/null
  python -c "
import json
d=json.load(open('$OUT/tmp.json'))
print('$cfg:', 'ms/step %.3f value %.4e' % (d['ms_per_step'], d['value']), {k: round(v['avg_ms'],3) for k,v in d['kernels'].items()})"
done | tee $OUT/secondary_configurations.txt
rm -f $OUT/tmp.json
timeout 400 python scripts/bench_lwfa_boosted.py > $OUT/lwfa_boosted.json 2> $OUT/lwfa_boosted.err; echo "config 5 rc=$?"
python -c "
import json
d=json.load(open('$OUT/lwfa_boosted.json'))
print('config 5: ms/step %.2f, %.3e particle-steps/s, %.3e cell-updates/s, particles %d -> %d' % (d['ms_per_step'], d['value'], d['cell_updates_per_s'], d['config']['particles_before'], d['config']['particles_after']))
for k,v in d['kernels'].items(): print('  %-18s %.3f ms per launch, %.2f launches per step, %.3f ms per step %s' % (k, v['avg_ms'], v['launches_per_step'], v['ms_per_step'], ('hbm %.3f' % v['hbm_frac']) if 'hbm_frac' in v else ''))
" | tee $OUT/lwfa_boosted.txt
