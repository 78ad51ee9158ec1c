#!/bin/bash
# A/B of an environment switch on the bench: bash scripts/gpu_ab_env.sh VAR "v1 v2 ..." [repeats]
set -u
OUT=$(pwd)/gpurun_out/ab
mkdir -p $OUT
export TMPDIR=/tmp
VAR=$1; VALUES=$2; REP=${3:-2}
for r in $(seq 1 $REP); do
  for V in $VALUES; do
    env $VAR=$V timeout 300 python bench.py --no-cpu-baseline --no-sanity > $OUT/bench_$V.json 2> $OUT/bench_$V.err
    python - $OUT/bench_$V.json "$VAR=$V" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
ph = {k: round(v["avg_ms"], 3) for k, v in (j.get("kernels") or {}).items()}
print(sys.argv[2], "ms/step %.3f" % j["ms_per_step"], "value %.3e" % j["value"], ph)
PY
  done
done
