#!/bin/bash
# Round 4: half tiles again on round 3's kernel (two workgroups of 79 KB per CU: variants 70 = 2 x 6 waves, 71 = 2 x 8 waves at
# 128 VGPRs) against production; parity of the two variants; the sort interval swept again on the production library.
#   gpurun --timeout 900 -- 'bash scripts/sessions/gpu_round4_h.sh'
set -u
OUT=$(pwd)/gpurun_out/r4h
mkdir -p $OUT
export TMPDIR=/tmp
WXA_EXTRA_DEFS=-DWXA_DEV_VARIANTS WXA_LIB_OUT=warpx_amd/libwarpx_amd_dev.so python -m warpx_amd.build --force > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
WXA_PRODUCT_LIB=$(pwd)/warpx_amd/libwarpx_amd_dev.so timeout 400 python scripts/variants.py base WXA_DEPOSIT_VARIANT=70 WXA_DEPOSIT_VARIANT=71 --repeat 3 2>&1 | grep -v "^\[{" | tail -10 | tee $OUT/deposit_half_tiles.txt
WXA_PRODUCT_LIB=$(pwd)/warpx_amd/libwarpx_amd_dev.so timeout 400 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "test_deposit_tile_variants and (70 or 71)" 2>&1 | tail -3 | tee $OUT/pytest_half_tiles.txt
for S in 3 4 5 6; do for r in 1 2; do timeout 200 python bench.py --no-cpu-baseline --no-sanity --steps 12 --sort-interval $S 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('sort interval $S', 'ms/step %.3f value %.3e' % (j['ms_per_step'], j['value']), {k: round(v['avg_ms'],3) for k,v in j['kernels'].items()})
"; done; done 2>&1 | tee $OUT/sort_interval_sweep.txt
du -sh $OUT
