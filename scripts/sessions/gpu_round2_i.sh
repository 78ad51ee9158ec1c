#!/bin/bash
# the boosted-frame tests on the MI355X and a fresh stamp of the PMC traffic for the final kernel sources
set -u
OUT=$(pwd)/gpurun_out/r2i
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_step_gpu.py -m gpu -q -rf -k "boosted or laser or lens or add_plasma or decks" 2>&1 | grep -E "passed|failed|FAILED|Error" | head -20 | tee $OUT/pytest.txt
timeout 900 python scripts/pmc_traffic.py $OUT/pmc > $OUT/pmc_traffic.log 2>&1
tail -12 $OUT/pmc_traffic.log
timeout 300 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python - $OUT/bench.json <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("ms/step %.3f value %.3e" % (j["ms_per_step"], j["value"]), {k: round(v["avg_ms"], 3) for k, v in j["kernels"].items()})
print(j["roofline"])
PY
rm -rf $OUT/pmc/*/*/*.db 2>/dev/null
du -sh $OUT
