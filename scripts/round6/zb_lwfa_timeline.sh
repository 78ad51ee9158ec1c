#!/bin/bash
# round 6, session zb: the dispatch timeline of BASELINE config 5 on one GPU (what runs, and what does not, between the phases the bench lists)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$(pwd)/gpurun_out/r6zb; mkdir -p $O; ROOTDIR=$(pwd)
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $O/prof -o trace -- \
    python $ROOTDIR/scripts/bench_lwfa_boosted.py --steps 6 ) > $O/rocprof.log 2>&1
f=$(find $O/prof -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python scripts/kernel_timeline.py $f 140 > $O/timeline_last_steps.txt
f2=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f2" ] && head -30 $f2 | cut -c1-200 > $O/kernel_stats_head.txt
f3=$(find $O/prof -name "*memory_copy_stats.csv" | head -1); [ -n "$f3" ] && cat $f3 > $O/memory_copy_stats.txt
rm -rf $O/prof
tail -50 $O/timeline_last_steps.txt
