#!/bin/bash
# Round 2, third GPU session: counters of the deposition variants (separate --pmc passes, no tracing), 128^3 x 8 ppc
# thermalised; then the A/B timing at 256^3 for the new lane spacing.
set -u
OUT=$(pwd)/gpurun_out/r2c
mkdir -p $OUT
export TMPDIR=/tmp
ROOTDIR=$(pwd)
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -rf -k "tile_variants" 2>&1 | tail -6 > $OUT/pytest_variants.txt
tail -3 $OUT/pytest_variants.txt
cd /tmp
PASSES=(
 "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
 "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU"
)
for V in ${VARIANTS:-0 4 8}; do
  i=0
  for P in "${PASSES[@]}"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $P --kernel-include-regex "deposit_tile" --output-format csv -d $OUT/v${V}_pass$i -o pmc -- \
        python $ROOTDIR/scripts/deposit_variants.py --ncell 128 --steps 3 --preroll 30 --variants $V > $OUT/v${V}_pass$i.log 2>&1
    echo "variant $V pass $i rc=$?"
  done
  python $ROOTDIR/scripts/summarize_pmc.py $OUT/v${V}_ "deposit_tile" > $OUT/v${V}_summary.txt 2>&1
  cat $OUT/v${V}_summary.txt
done
cd $ROOTDIR
timeout 600 python scripts/deposit_variants.py --variants ${TIMED:-0,4,8,9,6} > $OUT/variants.txt 2> $OUT/variants.err
cat $OUT/variants.txt | head -12
rm -rf $OUT/v*_pass*/*/*.db 2>/dev/null
du -sh $OUT
