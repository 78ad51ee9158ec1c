#!/bin/bash
# Round 4: sort interval re-swept with the stragglers written by tile (60 timed steps = a multiple of every interval)
set -u
OUT=$(pwd)/gpurun_out/r4r
mkdir -p $OUT
export TMPDIR=/tmp
for rep in 1 2; do
for si in 3 4 5 6; do
  timeout 300 python bench.py --steps 60 --warmup 6 --sort-interval $si --no-cpu-baseline --no-sanity > $OUT/bench_si${si}_$rep.json 2> $OUT/bench_si${si}_$rep.err
  python -c "import json;d=json.load(open('$OUT/bench_si${si}_$rep.json'));print('interval $si rep $rep', round(d['ms_per_step'],3), {k:round(v['avg_ms'],3) for k,v in d['kernels'].items()})"
done; done | tee $OUT/sort_interval_sweep.txt
