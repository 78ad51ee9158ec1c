#!/bin/bash
# Round 3, third GPU session: cooperative lane pairs (variants 20, 22) and 32-cell blocks (21, 22) in the deposition --
# parity, A/B timing, chunk clocks, SQ counters; clocks of the gather tile kernel.
set -u
OUT=$(pwd)/gpurun_out/r3c
mkdir -p $OUT
export TMPDIR=/tmp
ROOTDIR=$(pwd)
DEV=$ROOTDIR/warpx_amd/libwarpx_amd_dev.so
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "tile_variants or deposit_current_lds" 2>&1 | tail -4 > $OUT/pytest_kernels.txt
cat $OUT/pytest_kernels.txt
WXA_PRODUCT_LIB=$DEV timeout 600 python scripts/variants.py WXA_DEPOSIT_VARIANT=14 WXA_DEPOSIT_VARIANT=20 WXA_DEPOSIT_VARIANT=21 WXA_DEPOSIT_VARIANT=22 --repeat 3 \
    > $OUT/deposit_variants.txt 2> $OUT/deposit_variants.err
grep -v "^\[" $OUT/deposit_variants.txt | head -8; tail -2 $OUT/deposit_variants.err
timeout 300 python scripts/gather_profile.py > $OUT/gather_profile.txt 2>&1
tail -4 $OUT/gather_profile.txt
for V in -1 20 22; do
  timeout 300 python scripts/deposit_profile2.py $V > $OUT/deposit_phases_$V.txt 2>&1
  tail -10 $OUT/deposit_phases_$V.txt
done
cd /tmp
PASSES=(
 "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
 "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU"
)
i=0
for P in "${PASSES[@]}"; do
  i=$((i+1))
  WXA_PRODUCT_LIB=$DEV timeout 300 rocprofv3 --pmc $P --kernel-include-regex "deposit_tile_rows" --output-format csv -d $OUT/sq_pass$i -o pmc -- \
      python $ROOTDIR/scripts/variants.py WXA_DEPOSIT_VARIANT=22 --ncell 128 --steps 3 --preroll 30 --no-step-time > $OUT/sq_pass$i.log 2>&1
  echo "SQ pass $i rc=$?"
done
python $ROOTDIR/scripts/summarize_pmc.py $OUT/sq_ "deposit_tile_rows" > $OUT/sq_summary.txt 2>&1
cat $OUT/sq_summary.txt
cd $ROOTDIR
rm -rf $OUT/*/*/*.db $OUT/*/*.db 2>/dev/null
du -sh $OUT
