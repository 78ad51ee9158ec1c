#!/bin/bash
# Round 4: the secondary configurations at HEAD -- direct deposition (order 3 and 2), Vay pusher -- as bench lines.
#   gpurun --timeout 600 -- 'bash scripts/sessions/gpu_round4_l.sh'
set -u
OUT=$(pwd)/gpurun_out/${SESSION:-r4l}
mkdir -p $OUT
export TMPDIR=/tmp
for ARGS in "--deposition direct" "--deposition direct --order 2" "--pusher vay" "--order 2" "--order 1"; do
  timeout 200 python bench.py --no-cpu-baseline --no-sanity --steps 9 $ARGS 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$ARGS', 'ms/step %.3f value %.3e' % (j['ms_per_step'], j['value']), {k: round(v['avg_ms'],3) for k,v in j['kernels'].items()})
"; done 2>&1 | tee $OUT/secondary_configurations.txt
