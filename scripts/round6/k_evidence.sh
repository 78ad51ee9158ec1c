#!/bin/bash
# round 6, the evidence session at the round's last kernel sources: smoke, the whole -m gpu suite, the PMC passes (traffic + SQ,
# stamped), the default bench line as the driver runs it (with the CPU baseline), rocprofv3 --kernel-trace --stats of the same
# command, the step's timeline, the secondary configurations, BASELINE config 5 on one GPU, the product as 2 and 8 processes.
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$(pwd)/gpurun_out/r6fin; mkdir -p $O; ROOTDIR=$(pwd)
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v "^HIP version\|^ROCm version\|^Hostname\|^Librccl\|^RCCL" | tail -8 | tee $O/pytest_gpu.txt
timeout 1500 python scripts/pmc_traffic.py $O/pmc > $O/pmc_stdout.txt 2>&1; echo "pmc rc=$?"
cp $O/pmc/r6_pmc_counters.json $O/r6_pmc_counters.json && cp $O/pmc/r6_pmc_counters.json profiles/round6/r6_pmc_counters.json
rm -rf $O/pmc/FETCH_SIZE $O/pmc/WRITE_SIZE $O/pmc/sq1 $O/pmc/sq2
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
tail -c 1200 $O/bench.json
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o trace -- \
    python $ROOTDIR/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-phase-pass --no-sanity ) > $O/rocprof.log 2>&1
for f in $(find $O/prof -name "*kernel_stats*.csv" | head -1); do cp $f $O/kernel_stats.csv; head -14 $f; done
f=$(find $O/prof -name "*kernel_trace.csv" | head -1); python scripts/kernel_timeline.py $f 90 > $O/timeline_last_steps.txt
rm -rf $O/prof
for cfg in "--pusher vay" "--deposition direct" "--order 2" "--order 1" "--order 4" "--sort-interval 3"; do
  timeout 300 python bench.py --steps 12 --warmup 6 --no-cpu-baseline --no-sanity $cfg > $O/tmp.json 2>/dev/null
  python -c "
import json
d=json.load(open('$O/tmp.json'))
print('$cfg:', 'ms/step %.3f value %.4e' % (d['ms_per_step'], d['value']), {k: round(v['avg_ms'],3) for k,v in d['kernels'].items()})"
done | tee $O/secondary_configurations.txt
rm -f $O/tmp.json
timeout 500 python scripts/bench_lwfa_boosted.py > $O/lwfa_boosted.json 2> $O/lwfa_boosted.err; echo "config 5 rc=$?"
python -c "
import json
d=json.load(open('$O/lwfa_boosted.json'))
print('config 5: ms/step %.2f, %.3e particle-steps/s, %.3e cell-updates/s, particles %d -> %d' % (d['ms_per_step'], d['value'], d['cell_updates_per_s'], d['config']['particles_before'], d['config']['particles_after']))
for k,v in d['kernels'].items(): print('  %-18s %.3f ms per launch, %.2f launches per step, %.3f ms per step %s' % (k, v['avg_ms'], v['launches_per_step'], v['ms_per_step'], ('hbm %.3f' % v['hbm_frac']) if 'hbm_frac' in v else ''))
" | tee $O/lwfa_boosted.txt
for n in 2 8; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2970$n bench.py --gpus $n \
     --backend gloo --ranks-per-gpu $n --ncell 128 --steps 10 --warmup 3 --dry-comm 2>/dev/null | grep '^{' > $O/dry_comm_${n}_processes.json
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2971$n bench.py --gpus $n \
     --backend gloo --ranks-per-gpu $n --ncell 128 --steps 10 --warmup 3 2>/dev/null | grep '^{' > $O/bench_${n}_processes.json
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2972$n bench.py --gpus $n \
     --backend gloo --ranks-per-gpu $n --ncell 128 --steps 10 --warmup 3 --single-precision-comms 2>/dev/null | grep '^{' > $O/bench_${n}_processes_f32_wire.json
  python -c "
import json
for f in ('bench_${n}_processes.json','bench_${n}_processes_f32_wire.json'):
    d=json.load(open('$O/'+f)); print(f, 'ms/step %.2f' % d['ms_per_step'], 'MB sent per step %.2f' % d['exchange']['MB_sent_per_step'], 'sanity', d['sanity']['ok'])"
done | tee $O/processes.txt
