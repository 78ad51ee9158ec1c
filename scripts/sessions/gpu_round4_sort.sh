#!/bin/bash
# Round 4: the sort's scatter working the keys out again from the positions it loads (default) against reading them from the
# array the count pass wrote (WXA_SORT_SCATTER=1): the two sort kernels in the kernel trace of a short bench run each
set -u
OUT=$(pwd)/gpurun_out/r4sort
mkdir -p $OUT
export TMPDIR=/tmp
ROOTDIR=$(pwd)
for rep in 1 2; do
for v in recompute array; do
  if [ $v = array ]; then export WXA_SORT_SCATTER=1; else unset WXA_SORT_SCATTER; fi
  ( cd /tmp && WXA_PRODUCT_LIB=$ROOTDIR/warpx_amd/libwarpx_amd_dev.so timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$v -o trace -- \
    python $ROOTDIR/bench.py --steps 12 --no-cpu-baseline --no-phase-pass --no-sanity ) > $OUT/rocprof_$v.log 2>&1
  for f in $(find $OUT/prof_$v -name "*kernel_stats*.csv" | head -1); do echo "$v rep $rep: $(grep 'sort_scatter_window\|sort_count' $f | awk -F, '{printf "%s %.0f us  ", substr($1,12,22), $(NF-4)/1000}')"; done
  rm -rf $OUT/prof_$v
done; done | tee $OUT/sort_keys_recomputed.txt
timeout 300 python -m pytest tests/test_kernels_gpu.py tests/test_step_gpu.py -m gpu -q -k "sort or parity_in_the_benchmark or retired" 2>&1 | grep -E "passed|failed" | tee $OUT/pytest.txt
