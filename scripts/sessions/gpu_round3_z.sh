#!/bin/bash
# Round 3, last GPU sessions: order-4 particle shape on the MI355X (kernel and step parity), the bench line after the
# shape-factor generalisation, and (STAMP=1) the PMC traffic passes re-stamped with HEAD's kernel sources.
set -u
OUT=$(pwd)/gpurun_out/r3z
mkdir -p $OUT
export TMPDIR=/tmp
if [ "${STAMP:-0}" = "1" ]; then
  timeout 400 python scripts/pmc_traffic.py $OUT/pmc > $OUT/pmc_traffic.log 2>&1
  tail -12 $OUT/pmc_traffic.log
  rm -rf $OUT/pmc/*/*/*.db $OUT/pmc/*/*.db 2>/dev/null
else
  timeout 200 python -m pytest tests/test_kernels_gpu.py tests/test_step_gpu.py -m gpu -q -k "test_gather_push[ or test_deposit_current[ or test_esirkepov_continuity or test_deposit_charge or test_uniform_plasma_parity[" 2>&1 | tail -3 | tee $OUT/pytest_order4.txt
  timeout 200 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
  python -c "import json;d=json.load(open('$OUT/bench.json'));print(d['ms_per_step'],d['value'],{k:round(v['avg_ms'],3) for k,v in d['kernels'].items()})"
fi
