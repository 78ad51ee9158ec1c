#!/bin/bash
# Round 4: phase D of the deposition as six passes with compile-time (kind, component) -- the fix of the fp32 order-2 build,
# whose merged last ds_add_f32 had an undefined address on the jz edge -- and the gather's stragglers by tile as production:
# all kernel and step parity tests on the production library, A/B of the deposition against session j's configuration, bench.
set -u
OUT=$(pwd)/gpurun_out/r4q
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -X faulthandler -m pytest tests/test_kernels_gpu.py -m gpu -q -rf > $OUT/pytest_kernels_production.txt 2>&1
echo kernels rc=$?; tail -4 $OUT/pytest_kernels_production.txt | cut -c1-200
WXA_PRODUCT_LIB=$(pwd)/warpx_amd/libwarpx_amd_dev.so timeout 500 python scripts/variants.py base WXA_DEPOSIT_VARIANT=83 --repeat 3 2>&1 | grep -v "^\[{" | tail -8 | tee $OUT/deposit_six_passes_vs_j.txt
WXA_PRODUCT_LIB=$(pwd)/warpx_amd/libwarpx_amd_dev.so timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "variants" 2>&1 | tail -2 | tee $OUT/pytest_variants_dev.txt
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
python -c "import json;d=json.load(open('$OUT/bench.json'));print(d['ms_per_step'],d['value'],d['roofline'],{k:round(v['avg_ms'],3) for k,v in d['kernels'].items()})"; tail -3 $OUT/bench.err
timeout 900 python -X faulthandler -m pytest tests/test_step_gpu.py -m gpu -q -rf -k "not 256_electrons_and_protons" > $OUT/pytest_step_production.txt 2>&1
echo step rc=$?; tail -4 $OUT/pytest_step_production.txt | cut -c1-200
