#!/bin/bash
# Round 4, first GPU session.  Round 3's second session had no GPU minutes left: the reduced diagnostics
# (wxa_reduce_field / wxa_reduce_particles, host/ReducedDiags.hpp) and the one-plotfile-for-all-bricks writer have only
# run on the CPU execution model.  This session: (1) those tests first (seconds), (2) the default bench line + kernel
# trace, (3) the whole -m gpu suite, (4) smoke.  PMC=1 re-takes the FETCH / WRITE passes (only needed when a kernel source
# under warpx_amd/csrc/*.hip,*.hpp has changed: bench.py prints whether the committed stamp still matches).
#   gpurun --timeout 1500 -- 'bash scripts/sessions/gpu_round4_a.sh'
set -u
OUT=$(pwd)/gpurun_out/r4a
mkdir -p $OUT
export TMPDIR=/tmp
ROOTDIR=$(pwd)
timeout 300 python -m pytest tests/test_kernels_gpu.py tests/test_step_gpu.py tests/test_multibrick_gpu.py -m gpu -q -rf \
    -k "reduce_ or reduced_diags or one_plotfile or full_diagnostics" 2>&1 | tail -15 | tee $OUT/pytest_new_since_r3z.txt
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
python -c "import json;d=json.load(open('$OUT/bench.json'));print(d['ms_per_step'],d['value'],d['roofline']['traffic'],{k:round(v['avg_ms'],3) for k,v in d['kernels'].items()})"; tail -3 $OUT/bench.err
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- \
    python $ROOTDIR/bench.py --no-cpu-baseline --no-phase-pass --no-sanity ) > $OUT/rocprof.log 2>&1
tail -2 $OUT/rocprof.log
for f in $(find $OUT/prof -name "*kernel_stats*.csv" | head -1); do head -8 $f | cut -c1-160; cp $f $OUT/kernel_stats.csv; done
if [ "${PMC:-0}" = "1" ]; then
  timeout 600 python scripts/pmc_traffic.py $OUT/pmc > $OUT/pmc_traffic.log 2>&1
  tail -24 $OUT/pmc_traffic.log
fi
rm -rf $OUT/prof/*/*.db $OUT/pmc/*/*/*.db $OUT/prof/*.db $OUT/pmc/*/*.db $OUT/prof/*kernel_trace.csv 2>/dev/null
timeout ${PYTEST_LIMIT:-1150} python -m pytest tests -m gpu -q -rf --durations=8 2>&1 | tail -30 > $OUT/pytest_gpu.txt
tail -22 $OUT/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
du -sh $OUT
