#!/bin/bash
# round 6, session l: what the counting half of the folded sort would cost inside the deposition kernel (a probe build: key of
# the free-flight position, rank from an LDS histogram, one 8-byte store per particle; nothing uses the result)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$(pwd)/gpurun_out/r6l; mkdir -p $O
for rep in 1 2 3; do for v in "" "WXA_COUNT_PROBE=1"; do
  env WXA_PRODUCT_LIB=warpx_amd/libwarpx_amd_probe.so $v timeout 300 python bench.py --steps 12 --warmup 6 --no-cpu-baseline --no-sanity > $O/tmp.json 2>/dev/null
  python -c "
import json
d=json.load(open('$O/tmp.json'))
print('rep $rep [$v]: ms/step %.3f' % d['ms_per_step'], {k: round(v['avg_ms'],3) for k,v in d['kernels'].items() if k in ('GatherAndPush','CurrentDeposition')})"
done; done | tee $O/count_probe.txt
rm -f $O/tmp.json
