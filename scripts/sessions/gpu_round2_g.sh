#!/bin/bash
# Round 2, gather session: the new lens / boosted-frame tests, the 640-lane tile gather against the 512-lane one, and
# the SQ counters of the gather kernel (separate --pmc passes, no tracing), 128^3 x 8 ppc thermalised.
set -u
OUT=$(pwd)/gpurun_out/r2g
mkdir -p $OUT
export TMPDIR=/tmp
ROOTDIR=$(pwd)
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_step_gpu.py -m gpu -q -rf -k "lens or gather_push or two_parts or external" 2>&1 | tail -6 > $OUT/pytest.txt
cat $OUT/pytest.txt
for T in 0 640 0 640; do
  WXA_GATHER_THREADS=$T timeout 300 python bench.py --no-cpu-baseline --no-sanity > $OUT/bench_t$T.json 2> $OUT/bench_t$T.err
  python - $OUT/bench_t$T.json $T <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("gather threads", sys.argv[2], "ms/step", j["ms_per_step"], "value %.3e" % j["value"], j.get("phases_ms_per_step") or j.get("phases"))
PY
done
cd /tmp
PASSES=(
 "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
 "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU"
)
i=0
for P in "${PASSES[@]}"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $P --kernel-include-regex "gather_push_tile" --output-format csv -d $OUT/g_pass$i -o pmc -- \
      python $ROOTDIR/scripts/deposit_variants.py --ncell 128 --steps 3 --preroll 30 --variants 14 > $OUT/g_pass$i.log 2>&1
  echo "gather pmc pass $i rc=$?"
done
python $ROOTDIR/scripts/summarize_pmc.py $OUT/g_ "gather_push_tile" > $OUT/gather_pmc_summary.txt 2>&1
cat $OUT/gather_pmc_summary.txt
rm -rf $OUT/g_pass*/*/*.db 2>/dev/null
du -sh $OUT
