#!/bin/bash
# Round 4, fourth GPU session: where the time of the two particle kernels goes at HEAD -- SQ counters of the SHIPPED
# kernels (production library, no variant switch), and the timing experiments "no LDS atomics" / "no arithmetic".
#   gpurun --timeout 1200 -- 'bash scripts/sessions/gpu_round4_d.sh'
set -u
OUT=$(pwd)/gpurun_out/r4d
mkdir -p $OUT
export TMPDIR=/tmp
ROOTDIR=$(pwd)
cd /tmp
PASSES=(
 "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
 "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVE_CYCLES"
)
i=0
for P in "${PASSES[@]}"; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $P --kernel-trace --kernel-include-regex "deposit_tile_rows|gather_push_tile" --output-format csv -d $OUT/sq/pass$i -o pmc -- \
      python $ROOTDIR/bench.py --steps 6 --warmup 0 --no-cpu-baseline --no-phase-pass --no-sanity > $OUT/sq_pass$i.log 2>&1
  echo "SQ pass $i rc=$?"; tail -2 $OUT/sq_pass$i.log | cut -c1-300
done
cd $ROOTDIR
python scripts/sq_summary.py $OUT/sq "deposit_tile_rows|gather_push_tile" --last 6 2>&1 | tee $OUT/sq_summary_shipped_kernels.txt
rm -rf $OUT/sq/*/*/*.db $OUT/sq/*/*.db 2>/dev/null
WXA_EXTRA_DEFS=-DWXA_DEV_VARIANTS WXA_LIB_OUT=warpx_amd/libwarpx_amd_dev.so python -m warpx_amd.build --force > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
WXA_PRODUCT_LIB=$(pwd)/warpx_amd/libwarpx_amd_dev.so timeout 400 python scripts/variants.py base WXA_DEPOSIT_VARIANT=101 WXA_DEPOSIT_VARIANT=102 \
   WXA_GATHER_RB=2,WXA_GATHER_PF=7 WXA_GATHER_RB=2,WXA_GATHER_PF=9 --repeat 2 2>&1 | grep -v "^\[{" | tail -12 | tee $OUT/timing_experiments.txt
du -sh $OUT
