#!/bin/bash
# phase clocks of the new deposition kernels + A/B timing with the 16-wide LDS rows
set -u
OUT=$(pwd)/gpurun_out/r2f
mkdir -p $OUT
export TMPDIR=/tmp
for V in ${PROFV:-14 16 12}; do
  timeout 300 python scripts/deposit_profile2.py $V > $OUT/prof_v$V.txt 2>&1; cat $OUT/prof_v$V.txt | tail -9
done
timeout 600 python scripts/deposit_variants.py --variants ${TIMED:-0,12,14,16,17} > $OUT/variants.txt 2> $OUT/variants.err
grep "^variant" $OUT/variants.txt
