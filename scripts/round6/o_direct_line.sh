#!/bin/bash
# round 6, session o: the direct-deposition line with the non-Galerkin gather's stage sized per component (73 KB: two workgroups per CU)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$(pwd)/gpurun_out/r6o; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "gather_push or uniform_plasma_parity or picmi or direct or bricks_on_one_gpu" 2>&1 | grep -v "^HIP version\|^ROCm version\|^Hostname\|^Librccl\|^RCCL" | tail -3 | tee $O/pytest.txt
for cfg in "--deposition direct" "--deposition direct --pusher vay" "--deposition direct --order 2" ""; do
  timeout 300 python bench.py --steps 12 --warmup 6 --no-cpu-baseline --no-sanity $cfg > $O/tmp.json 2>/dev/null
  python -c "
import json
d=json.load(open('$O/tmp.json'))
print('[$cfg]:', 'ms/step %.3f value %.4e' % (d['ms_per_step'], d['value']), {k: round(v['avg_ms'],3) for k,v in d['kernels'].items() if k in ('GatherAndPush','CurrentDeposition')})"
done | tee $O/direct_line.txt
rm -f $O/tmp.json
