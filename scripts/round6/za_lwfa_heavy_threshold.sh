#!/bin/bash
# round 6, session za: with the tiles in turn over the XCDs, from how many particles a tile is shared by several workgroups
# (WXA_HEAVY_TILE, default 8192 for streaming plasmas) and from how many lanes on one frame a wave sums first (WXA_WAVE_SUM_MIN, 16)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$(pwd)/gpurun_out/r6za; mkdir -p $O
for v in ${SWEEP:-"WXA_HEAVY_TILE=8192" "WXA_HEAVY_TILE=4096" "WXA_HEAVY_TILE=6144" "WXA_HEAVY_TILE=16384" "WXA_WAVE_SUM_MIN=8" "WXA_WAVE_SUM_MIN=32" "WXA_HEAVY_TILE=8192"}; do
  env $v timeout 500 python scripts/bench_lwfa_boosted.py --steps 30 > $O/tmp.json 2>/dev/null
  python -c "
import json
d=json.load(open('$O/tmp.json'))
print('$v: ms/step %.2f' % d['ms_per_step'], {k: round(v['avg_ms'],2) for k,v in d['kernels'].items() if k in ('CurrentDeposition','GatherAndPush','Redistribute')})"
done | tee $O/heavy_threshold.txt
rm -f $O/tmp.json
