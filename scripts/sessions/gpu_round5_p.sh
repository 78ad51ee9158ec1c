#!/bin/bash
# Round 5, session p: where config 5's deposition time goes -- the particles per cell and tile of the filled window, and the
# kernels of the last steps one by one (rocprofv3 kernel trace).
set -u
OUT=$(pwd)/gpurun_out/r5p
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python scripts/round5/lwfa_population_probe.py > $OUT/population.txt 2> $OUT/population.err; echo "probe rc=$?"
cat $OUT/population.txt
cd /tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python $GRAFT_REPO_ROOT/scripts/bench_lwfa_boosted.py --steps 6 > $OUT/trace_bench.json 2> $OUT/trace_bench.err; echo "trace rc=$?"
cd $GRAFT_REPO_ROOT
f=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python scripts/kernel_timeline.py $f 400 > $OUT/timeline_last_steps.txt
python - <<PY
import csv, collections
rows = []
with open("$f") as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:70]))
rows.sort()
rows = rows[-600:]
acc = collections.defaultdict(lambda: [0, 0.0])
for s, e, k in rows:
    acc[k][0] += 1; acc[k][1] += (e - s) / 1e6
for k, (c, ms) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:14]:
    print("%8.2f ms %5d calls %8.3f ms each  %s" % (ms, c, ms / c, k))
PY
rm -rf $OUT/trace
