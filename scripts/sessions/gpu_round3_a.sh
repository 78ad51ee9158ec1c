#!/bin/bash
# Round 3, first GPU session: parity of the two changed kernels, A/B of the gather's LDS read pipeline, phase clocks and SQ
# counters of the deposition kernel at HEAD, SQ counters of the gather at HEAD, FETCH/WRITE calibration.
set -u
OUT=$(pwd)/gpurun_out/r3a
mkdir -p $OUT
export TMPDIR=/tmp
ROOTDIR=$(pwd)
DEV=$ROOTDIR/warpx_amd/libwarpx_amd_dev.so
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "gather or deposit" 2>&1 | tail -4 > $OUT/pytest_kernels.txt
cat $OUT/pytest_kernels.txt
WXA_PRODUCT_LIB=$DEV timeout 600 python scripts/variants.py base WXA_GATHER_RB=0 WXA_GATHER_RB=1 WXA_GATHER_RB=2 WXA_GATHER_RB=3 --repeat 2 \
    > $OUT/gather_variants.txt 2> $OUT/gather_variants.err
grep -v "^\[" $OUT/gather_variants.txt | head -12; tail -2 $OUT/gather_variants.err
timeout 300 python scripts/deposit_profile2.py -1 > $OUT/deposit_phases.txt 2>&1
tail -9 $OUT/deposit_phases.txt
cd /tmp
PASSES=(
 "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
 "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU"
)
i=0
for P in "${PASSES[@]}"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $P --kernel-include-regex "deposit_tile_rows|gather_push_tile" --output-format csv -d $OUT/sq_pass$i -o pmc -- \
      python $ROOTDIR/scripts/variants.py base --ncell 128 --steps 3 --preroll 30 --no-step-time > $OUT/sq_pass$i.log 2>&1
  echo "SQ pass $i rc=$?"
done
python $ROOTDIR/scripts/summarize_pmc.py $OUT/sq_ "deposit_tile_rows|gather_push_tile" > $OUT/sq_summary.txt 2>&1
cat $OUT/sq_summary.txt
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 120 rocprofv3 --pmc $C --output-format csv -d $OUT/calib_$C -o pmc -- $ROOTDIR/scripts/microbench/fetch_calib > $OUT/calib_$C.log 2>&1
  echo "calib $C rc=$?"
done
python - $OUT <<'PY'
import csv, glob, sys
from collections import defaultdict
out = sys.argv[1]
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    per = defaultdict(lambda: defaultdict(float))
    for f in glob.glob(f"{out}/calib_{c}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == c:
                per[int(row["Dispatch_Id"])][row["Kernel_Name"].split("(")[0]] += float(row["Counter_Value"])
    for d in sorted(per):
        for k, v in per[d].items():
            print(f"{c} dispatch {d:3d} {k:14s} {v:14.1f} KiB")
PY
grep known $OUT/calib_FETCH_SIZE.log | head -12
cd $ROOTDIR
rm -rf $OUT/*/*/*.db $OUT/*/*.db 2>/dev/null
du -sh $OUT
