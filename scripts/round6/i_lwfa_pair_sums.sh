#!/bin/bash
# round 6, session i: BASELINE config 5 on one GPU with the streaming deposition's lane pairs summed before the LDS atomic
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$(pwd)/gpurun_out/r6i; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "streaming or boosted or btd or nci" 2>&1 | tail -3 | tee $O/pytest_streaming.txt
for v in "" "WXA_STREAM_WAVES_8=1"; do
env $v timeout 600 python scripts/bench_lwfa_boosted.py > $O/lwfa.json 2> $O/lwfa.err; echo "rc=$?"
python -c "
import json
d=json.load(open('$O/lwfa.json'))
print('config 5 [$v]: ms/step %.2f, %.3e particle-steps/s, particles %d -> %d' % (d['ms_per_step'], d['value'], d['config']['particles_before'], d['config']['particles_after']))
for k,v in d['kernels'].items(): print('  %-18s %.3f ms per launch, %.2f launches per step, %.3f ms per step' % (k, v['avg_ms'], v['launches_per_step'], v['ms_per_step']))
"
done | tee $O/lwfa_boosted.txt
