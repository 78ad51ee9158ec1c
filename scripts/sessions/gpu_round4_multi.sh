#!/bin/bash
# Round 4: the periodic guard fill of E and B and the periodic sum of J as one launch per direction for all fields
set -u
OUT=$(pwd)/gpurun_out/r4multi
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_step_gpu.py tests/test_multibrick_gpu.py -m gpu -q -rf -k "boundary or parity or golden or multibrick or bricks" 2>&1 | tail -4 | tee $OUT/pytest.txt
for rep in 1 2; do
timeout 300 python bench.py --steps 20 --no-cpu-baseline --no-sanity > $OUT/bench_$rep.json 2> $OUT/bench_$rep.err
python -c "import json;d=json.load(open('$OUT/bench_$rep.json'));print('rep $rep', round(d['ms_per_step'],3), {k:round(v['avg_ms'],3) for k,v in d['kernels'].items()})"
done | tee $OUT/bench.txt
