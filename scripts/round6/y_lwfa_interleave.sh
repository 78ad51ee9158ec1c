#!/bin/bash
# round 6, session y: the tiles of a streaming plasma handed to the workgroups in turn (consecutive tiles on different XCDs) against a
# contiguous eighth of the tiles per XCD: BASELINE config 5 on one GPU
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$(pwd)/gpurun_out/r6y; mkdir -p $O
for v in 1 0 1 0; do
  WXA_TILE_INTERLEAVE=$v timeout 500 python scripts/bench_lwfa_boosted.py --steps 30 > $O/tmp.json 2>/dev/null
  python -c "
import json
d=json.load(open('$O/tmp.json'))
print('WXA_TILE_INTERLEAVE=$v: ms/step %.2f' % d['ms_per_step'], {k: round(v['avg_ms'],2) for k,v in d['kernels'].items() if k in ('CurrentDeposition','GatherAndPush','Redistribute')})"
done | tee $O/interleave.txt
rm -f $O/tmp.json
