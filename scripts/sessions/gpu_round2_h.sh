#!/bin/bash
# new tests of the boosted-frame / lens work on the MI355X, and the sort interval revisited with this round's kernels
set -u
OUT=$(pwd)/gpurun_out/r2h
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_step_gpu.py -m gpu -q -rf -k "lens or add_plasma or boosted or decks or laser" 2>&1 | tail -6 > $OUT/pytest.txt
cat $OUT/pytest.txt
for S in 2 3 4 5 6; do
  timeout 300 python bench.py --no-cpu-baseline --no-sanity --sort-interval $S > $OUT/bench_s$S.json 2> $OUT/bench_s$S.err
  python - $OUT/bench_s$S.json $S <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
ph = {k: round(v["avg_ms"], 3) for k, v in (j.get("kernels") or {}).items()}
print("sort interval", sys.argv[2], "ms/step %.3f" % j["ms_per_step"], "value %.3e" % j["value"], ph)
PY
done
