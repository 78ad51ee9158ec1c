#!/bin/bash
# Round 5, session d: the whole -m gpu suite on the folded sort; gather stragglers on 2048 workgroups against 512 (second
# library); the complete kernel timeline of three steps (idle gaps between dispatches: what a graph of the step could win).
set -u
OUT=$(pwd)/gpurun_out/r5d
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -4 | tee $OUT/pytest_gpu.txt
line() { python -c "
import json,sys
d=json.load(open('$1'))
print('$2', 'ms/step %.3f value %.4e' % (d['ms_per_step'], d['value']), {k: round(v['avg_ms'],3) for k,v in d['kernels'].items()})"; }
for rep in 1 2; do
  WXA_PRODUCT_LIB=$(pwd)/warpx_amd/libwarpx_amd_s512.so timeout 300 python bench.py --steps 24 --warmup 6 --no-cpu-baseline --no-sanity > $OUT/s512_$rep.json 2> $OUT/s512_$rep.err
  line $OUT/s512_$rep.json "stragglers on 512 workgroups rep $rep"
  timeout 300 python bench.py --steps 24 --warmup 6 --no-cpu-baseline --no-sanity > $OUT/s2048_$rep.json 2> $OUT/s2048_$rep.err
  line $OUT/s2048_$rep.json "stragglers on 2048 workgroups rep $rep"
done 2>&1 | tee $OUT/ab.txt
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -o trace -- python $OLDPWD/bench.py --steps 6 --warmup 3 --preroll 12 --no-cpu-baseline --no-sanity --no-phase-pass ) > $OUT/rocprof.log 2>&1
f=$(find $OUT/prof -name "*kernel_trace.csv" | head -1)
python scripts/kernel_timeline.py $f 260 > $OUT/timeline_last_steps.txt
tail -3 $OUT/timeline_last_steps.txt
python scripts/kernel_durations.py $f "stragglers" | tail -12
rm -rf $OUT/prof
