#!/bin/bash
# Round 5, session s: the SQ counters of config 5's deposition kernel (last launches of a filled window), the z-march
# Godfrey filter on the hardware (test + config 5 line).
set -u
OUT=$(pwd)/gpurun_out/r5s
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "filter" 2>&1 | grep -v "^HIP version\|^ROCm version\|^Hostname\|^Librccl\|^RCCL" | tail -2 | tee $OUT/pytest_filter.txt
timeout 600 python scripts/bench_lwfa_boosted.py > $OUT/lwfa_boosted.json 2> $OUT/lwfa_boosted.err; echo "line rc=$?"
python -c "
import json
d=json.load(open('$OUT/lwfa_boosted.json'))
print('config 5: ms/step %.2f, %.3e particle-steps/s, %.3e cell-updates/s, particles %d -> %d' % (d['ms_per_step'], d['value'], d['cell_updates_per_s'], d['config']['particles_before'], d['config']['particles_after']))
for k,v in d['kernels'].items(): print('  %-18s %.3f ms per launch, %.2f launches per step, %.3f ms per step %s' % (k, v['avg_ms'], v['launches_per_step'], v['ms_per_step'], ('hbm %.3f' % v['hbm_frac']) if 'hbm_frac' in v else ''))
" | tee $OUT/lwfa_boosted.txt
cd /tmp
timeout 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-include-regex "deposit_tile_rows" --output-format csv -d $OUT/pmc -o pmc -- python $GRAFT_REPO_ROOT/scripts/bench_lwfa_boosted.py --steps 4 > $OUT/pmc_bench.json 2> $OUT/pmc_bench.err; echo "pmc rc=$?"
cd $GRAFT_REPO_ROOT
f=$(find $OUT/pmc -name "*counter_collection.csv" | head -1)
python - <<PY | tee $OUT/deposit_sq_counters.txt
import csv, collections
rows = collections.defaultdict(dict)
order = []
with open("$f") as fh:
    for r in csv.DictReader(fh):
        d = int(r["Dispatch_Id"])
        if d not in rows: order.append(d)
        rows[d][r["Counter_Name"]] = rows[d].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
last = order[-6:]
v = {k: sum(rows[d].get(k, 0.0) for d in last) / len(last) for k in rows[last[0]]}
cyc = v["SQ_BUSY_CYCLES"] / 32
print("mean of the last %d dispatches of deposit_tile_rows_kernel (of %d)" % (len(last), len(order)))
for k in sorted(v): print("  %-24s %.4e" % (k, v[k]))
print("cycles per launch %.3e (%.2f ms at 2.09 GHz)" % (cyc, cyc / 2.09e6))
print("valu_busy_frac %.3f" % (4.0 * v["SQ_ACTIVE_INST_VALU"] / 1024 / cyc))
print("lds_array_busy_frac %.3f" % (v["SQ_LDS_IDX_ACTIVE"] / 256 / cyc))
print("lds_conflict_frac %.3f" % ((v["SQ_LDS_BANK_CONFLICT"] + v["SQ_LDS_ADDR_CONFLICT"]) / v["SQ_LDS_IDX_ACTIVE"]))
print("lds_array_cycles_per_lds_instruction %.2f" % (v["SQ_LDS_IDX_ACTIVE"] / v["SQ_INSTS_LDS"]))
print("VALU instructions per LDS instruction %.2f" % (v["SQ_INSTS_VALU"] / v["SQ_INSTS_LDS"]))
PY
rm -rf $OUT/pmc
