#!/bin/bash
# Round 5, session c: own / foreign slots per cell (no second pass over the record, no id load without retired particles)
# -- parity, then A/B against the sort as passes of its own at sort intervals 2, 3, 4.
set -u
OUT=$(pwd)/gpurun_out/r5c
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "sort_folded or gather_push or appended_tail" 2>&1 | tail -3 | tee $OUT/pytest_kernels.txt
timeout 600 python -m pytest tests/test_step_gpu.py -m gpu -q -k "test_uniform_plasma_parity" 2>&1 | tail -3 | tee $OUT/pytest_step.txt
line() { python -c "
import json,sys
d=json.load(open('$1'))
print('$2', 'ms/step %.3f value %.4e' % (d['ms_per_step'], d['value']), {k: round(v['avg_ms'],3) for k,v in d['kernels'].items()})"; }
for rep in 1 2; do
  for si in 3; do
    WXA_SORT_IN_PUSH=0 timeout 300 python bench.py --steps 24 --warmup 6 --sort-interval $si --no-cpu-baseline --no-sanity > $OUT/classic_si${si}_$rep.json 2> $OUT/classic_si${si}_$rep.err
    line $OUT/classic_si${si}_$rep.json "sort as passes of its own si$si rep $rep"
  done
  for si in 2 3 4; do
    timeout 300 python bench.py --steps 24 --warmup 6 --sort-interval $si --no-cpu-baseline --no-sanity > $OUT/fold_si${si}_$rep.json 2> $OUT/fold_si${si}_$rep.err
    line $OUT/fold_si${si}_$rep.json "sort in the push si$si rep $rep"
  done
done 2>&1 | tee $OUT/ab.txt
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -o trace -- python $OLDPWD/bench.py --steps 6 --warmup 3 --preroll 12 --sort-interval 3 --no-cpu-baseline --no-sanity --no-phase-pass ) > $OUT/rocprof.log 2>&1
f=$(find $OUT/prof -name "*kernel_trace.csv" | head -1)
python scripts/kernel_durations.py $f "gather_push|deposit_tile_rows|sort_|DeviceScan|rocprim|wrap|periodic" | tail -48 > $OUT/dispatches_si3.txt
cat $OUT/dispatches_si3.txt
rm -rf $OUT/prof
