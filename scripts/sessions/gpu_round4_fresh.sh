#!/bin/bash
# Round 4: the particle kernels on a fresh sort -- sort interval 1 (the deposition always sees the sort of its own step, the
# gather the previous step's) and 2, next to the production interval 3: what staleness costs each kernel
set -u
OUT=$(pwd)/gpurun_out/r4fresh
mkdir -p $OUT
export TMPDIR=/tmp
for si in 1 2 3; do
  timeout 300 python bench.py --steps 12 --warmup 6 --sort-interval $si --no-cpu-baseline --no-sanity > $OUT/bench_si$si.json 2> $OUT/bench_si$si.err
  python -c "import json;d=json.load(open('$OUT/bench_si$si.json'));print('interval $si', round(d['ms_per_step'],3), {k:round(v['avg_ms'],3) for k,v in d['kernels'].items()})"
done | tee $OUT/sort_interval_1_2_3.txt
