#!/bin/bash
# direct deposition on the rows kernel: parity tests, timing against the staged kernel of round 1, PMC traffic re-stamped
set -u
OUT=$(pwd)/gpurun_out/r2k
mkdir -p $OUT
export TMPDIR=/tmp
timeout 240 python -m pytest tests/test_kernels_gpu.py tests/test_step_gpu.py -m gpu -q -rf -k "deposit_current or tile_variants or uniform_plasma_parity or picmi or langmuir_golden" 2>&1 | grep -E "passed|failed|FAILED" | tee $OUT/pytest.txt
for V in -1 0; do
  WXA_DEPOSIT_VARIANT=$V timeout 120 python bench.py --no-cpu-baseline --no-sanity --deposition direct 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('direct deposition, variant $V:', 'ms/step %.3f value %.3e' % (j['ms_per_step'], j['value']), {k: round(v['avg_ms'],3) for k,v in j['kernels'].items()})
" | tee -a $OUT/direct.txt
done
timeout 200 python scripts/pmc_traffic.py $OUT/pmc > $OUT/pmc_traffic.log 2>&1
grep -E "sources_sha16" $OUT/pmc_traffic.log
rm -rf $OUT/pmc/*/*/*.db 2>/dev/null
