#!/bin/bash
# loads-before-stores in the sort scatter, the periodic wrap and the stencils' read-modify-writes: tests, stencil and CKC
# timings back to back, the bench line
set -u
OUT=$(pwd)/gpurun_out/r2j
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -rf -k "evolve or ckc or sort or periodic or partition or stencil or filter" 2>&1 | grep -E "passed|failed|FAILED" | tee $OUT/pytest.txt
timeout 300 python scripts/stencil_variants.py 256 30 0,3 2>&1 | grep -E "^round" | tee $OUT/stencils.txt
timeout 300 python scripts/ckc_timing.py 2>&1 | grep -E "plain|variant 1|Yee" | tee $OUT/ckc.txt
for r in 1 2; do
timeout 300 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python - $OUT/bench.json <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("ms/step %.3f value %.3e" % (j["ms_per_step"], j["value"]), {k: round(v["avg_ms"], 3) for k, v in j["kernels"].items()})
print({k: (round(v.get("hbm_frac", 0), 3), round(v.get("hbm_frac_back_to_back", 0), 3)) for k, v in j["kernels"].items() if k.startswith("Evolve")})
PY
done
