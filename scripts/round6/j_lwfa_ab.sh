#!/bin/bash
# round 6, session j: config 5 -- pair sums on / off, sort folded into the push (merged, every step) or the classic sort behind the window shift
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$(pwd)/gpurun_out/r6j; mkdir -p $O
for v in "WXA_SORT_MERGED=0" "WXA_SORT_MERGED=0 WXA_WAVE_SUM_MIN=63" "WXA_SORT_MERGED=1" "WXA_SORT_MERGED=0 WXA_STREAM_WAVES_8=1"; do
env $v timeout 600 python scripts/bench_lwfa_boosted.py > $O/lwfa.json 2> $O/lwfa.err; echo "rc=$?"
python -c "
import json
d=json.load(open('$O/lwfa.json'))
print('config 5 [$v]: ms/step %.2f, %.3e particle-steps/s' % (d['ms_per_step'], d['value']), {k: round(v['ms_per_step'],2) for k,v in d['kernels'].items()})
"
done | tee $O/lwfa_boosted_ab.txt
