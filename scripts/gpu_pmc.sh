#!/bin/bash
# PMC counter passes (separate from kernel-trace runs) for one bench configuration.
# Usage: bash scripts/gpu_pmc.sh "<bench args>" "<kernel regex>"
set -u
ARGS=${1:---ncell 128 --steps 2 --warmup 1 --no-cpu-baseline --no-phase-pass}
KREGEX=${2:-deposit}
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
ROOTDIR=$(pwd)
cd /tmp
rocprofv3 -L > $ROOTDIR/gpurun_out/pmc/counters_list.txt 2>&1
PASSES=(
 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS"
 "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"
 "FETCH_SIZE"
 "WRITE_SIZE"
 "TCC_HIT_sum TCC_MISS_sum"
)
i=0
for P in "${PASSES[@]}"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $P --output-format csv -d $ROOTDIR/gpurun_out/pmc/pass$i -o pmc -- \
      python $ROOTDIR/bench.py $ARGS > $ROOTDIR/gpurun_out/pmc/pass$i.log 2>&1
  echo "pass $i ($P) rc=$?"
done
cd $ROOTDIR
python scripts/summarize_pmc.py gpurun_out/pmc "$KREGEX" | tee gpurun_out/pmc/summary.txt
