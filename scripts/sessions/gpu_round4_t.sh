#!/bin/bash
# Round 4: direct deposition with the lane's two particles on one frame per component (240 LDS atomics per pair instead
# of 384) and the ones without a partner deferred, against the sequential version (dev variant 85); parity on the hardware.
set -u
OUT=$(pwd)/gpurun_out/r4t
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -X faulthandler -m pytest tests/test_kernels_gpu.py -m gpu -q -rf -k "deposit" 2>&1 | tail -4 | tee $OUT/pytest_deposit.txt
for rep in 1 2; do
  for v in 0 85; do
    WXA_DEPOSIT_VARIANT=$v WXA_PRODUCT_LIB=$(pwd)/warpx_amd/libwarpx_amd_dev.so timeout 300 python bench.py --deposition direct --steps 12 --no-cpu-baseline --no-sanity > $OUT/bench_direct_v${v}_$rep.json 2> $OUT/bench_direct_v${v}_$rep.err
    python -c "import json;d=json.load(open('$OUT/bench_direct_v${v}_$rep.json'));print('variant $v rep $rep', round(d['ms_per_step'],3), {k:round(v['avg_ms'],3) for k,v in d['kernels'].items()})"
  done
done | tee $OUT/direct_pair_frames.txt
timeout 600 python -m pytest tests/test_step_gpu.py -m gpu -q -rf -k "picmi or direct or golden" 2>&1 | tail -4 | tee $OUT/pytest_step_direct.txt
