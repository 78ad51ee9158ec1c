#!/bin/bash
# round 6, session zc: the uniform plasma with its tiles in turn over the XCDs against the contiguous eighth per XCD it ships with
# (WXA_HEAVY_TILE=100000 switches the unit table on without a heavy tile; WXA_TILE_INTERLEAVE picks the order)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$(pwd)/gpurun_out/r6zc; mkdir -p $O
for v in "WXA_TILE_INTERLEAVE=0" "WXA_TILE_INTERLEAVE=1" "WXA_TILE_INTERLEAVE=0" "WXA_TILE_INTERLEAVE=1"; do
  env WXA_HEAVY_TILE=100000 $v timeout 300 python bench.py --steps 12 --warmup 6 --no-cpu-baseline --no-sanity > $O/tmp.json 2>/dev/null
  python -c "
import json
d=json.load(open('$O/tmp.json'))
print('$v:', 'ms/step %.3f' % d['ms_per_step'], {k: round(v['avg_ms'],3) for k,v in d['kernels'].items() if k in ('GatherAndPush','CurrentDeposition')})"
done | tee $O/headline_interleave.txt
rm -f $O/tmp.json
