#!/bin/bash
# Round 5, session a: the cell sort folded into the push (csrc/push_sort.hpp) -- parity on the hardware, then the bench
# line against HEAD~'s library in the same session (base = the library built from the sources before the change).
set -u
OUT=$(pwd)/gpurun_out/r5a
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "sort_folded or gather_push or enforce_periodic or appended_tail" 2>&1 | tail -3 | tee $OUT/pytest_kernels.txt
timeout 600 python -m pytest tests/test_step_gpu.py -m gpu -q -k "test_uniform_plasma_parity" 2>&1 | tail -3 | tee $OUT/pytest_step.txt
line() { python -c "
import json,sys
d=json.load(open('$1'))
print('$2', 'ms/step %.3f value %.4e' % (d['ms_per_step'], d['value']), {k: round(v['avg_ms'],3) for k,v in d['kernels'].items()})"; }
for rep in 1 2; do
  WXA_PRODUCT_LIB=$(pwd)/warpx_amd/libwarpx_amd_base.so timeout 300 python bench.py --steps 12 --warmup 6 --no-cpu-baseline --no-sanity > $OUT/base_$rep.json 2> $OUT/base_$rep.err
  line $OUT/base_$rep.json "base si3 rep $rep"
  WXA_SORT_IN_PUSH=0 timeout 300 python bench.py --steps 12 --warmup 6 --no-cpu-baseline --no-sanity > $OUT/new_classic_$rep.json 2> $OUT/new_classic_$rep.err
  line $OUT/new_classic_$rep.json "new, sort as passes of its own si3 rep $rep"
  for si in 1 2 3 4; do
    timeout 300 python bench.py --steps 12 --warmup 6 --sort-interval $si --no-cpu-baseline --no-sanity > $OUT/fold_si${si}_$rep.json 2> $OUT/fold_si${si}_$rep.err
    line $OUT/fold_si${si}_$rep.json "sort in the push si$si rep $rep"
  done
done 2>&1 | tee $OUT/ab.txt
# the kernels of one run per dispatch (which push is the counting one, which the scattering one)
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -o trace -- python $OLDPWD/bench.py --steps 6 --warmup 3 --preroll 6 --sort-interval 2 --no-cpu-baseline --no-sanity --no-phase-pass ) > $OUT/rocprof.log 2>&1
f=$(find $OUT/prof -name "*kernel_trace.csv" | head -1)
python scripts/kernel_durations.py $f "gather_push|deposit_tile_rows|sort_|DeviceScan|wrap|periodic" | tail -60 > $OUT/dispatches_si2.txt
tail -40 $OUT/dispatches_si2.txt
rm -rf $OUT/prof
