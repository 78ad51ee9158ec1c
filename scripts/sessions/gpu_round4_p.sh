#!/bin/bash
set -u
OUT=$(pwd)/gpurun_out/r4p
mkdir -p $OUT
export TMPDIR=/tmp
for x in x1 x2; do
  WXA_PRODUCT_LIB=$(pwd)/warpx_amd/libwarpx_amd_$x.so timeout 300 python -X faulthandler -m pytest tests/test_kernels_gpu.py -m gpu -v -k "fp32_tiles" > $OUT/f32_$x.txt 2>&1
  echo $x rc=$?
  grep -n "FAILED\|PASSED\|Fatal\|passed\|failed" $OUT/f32_$x.txt | head -12 | cut -c1-160
done
