#!/bin/bash
# round 6, session b: the new tests of this round on the MI355X, the dispatch timeline of the step at HEAD, the deposition's phase clocks
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$(pwd)/gpurun_out/r6b; mkdir -p $O; ROOTDIR=$(pwd)
timeout 900 python -m pytest tests -m gpu -x -q -k "pack_unpack or single_precision_comms or baseline_config_1" 2>&1 | tail -4 | tee $O/pytest_new.txt
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof -o trace -- \
    python $ROOTDIR/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-phase-pass ) > $O/rocprof.log 2>&1
f=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python scripts/kernel_timeline.py $f 260 > $O/timeline_last_steps.txt; tail -3 $O/timeline_last_steps.txt
rm -rf $O/prof
timeout 600 python scripts/deposit_profile2.py -1 > $O/deposit_phase_clocks.txt 2>&1; cat $O/deposit_phase_clocks.txt | tail -12
