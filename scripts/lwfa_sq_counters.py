"""SQ counters of the streaming deposition kernel in BASELINE config 5 on one GPU (scripts/bench_lwfa_boosted.py), two rocprofv3
--pmc passes, mean of the last dispatches:   python scripts/lwfa_sq_counters.py [outdir]"""
import csv
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.abspath(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/lwfa_sq")
N_SE, N_CU, N_SIMD = 32, 256, 1024
PASSES = [
    "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES",
    "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_IFETCH",
]
v = {}
for ip, ctrs in enumerate(PASSES):
    d = os.path.join(out, f"sq{ip}")
    cmd = ["rocprofv3", "--pmc"] + ctrs.split() + ["--kernel-include-regex", "deposit_tile_rows", "--output-format", "csv", "-d", d, "-o", "pmc",
                                                   "--", sys.executable, os.path.join(ROOT, "scripts", "bench_lwfa_boosted.py"), "--steps", "4"]
    subprocess.run(cmd, cwd="/tmp", stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=dict(os.environ, TMPDIR="/tmp"), timeout=1500)
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    ids = sorted({int(r["Dispatch_Id"]) for r in rows})[-6:]
    for r in rows:
        if int(r["Dispatch_Id"]) in ids:
            v[r["Counter_Name"]] = v.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"]) / len(ids)
cyc = v["SQ_BUSY_CYCLES"] / N_SE
print("mean of the last 6 launches of the streaming deposition kernel:")
for k in sorted(v):
    print(f"  {k:24s} {v[k]:.4e}")
print(f"launch: {cyc:.3e} cycles")
print(f"VALU busy {4.0 * v['SQ_ACTIVE_INST_VALU'] / N_SIMD / cyc:.3f}, LDS array busy {v['SQ_LDS_IDX_ACTIVE'] / N_CU / cyc:.3f}, "
      f"conflicts {(v['SQ_LDS_BANK_CONFLICT'] + v['SQ_LDS_ADDR_CONFLICT']) / v['SQ_LDS_IDX_ACTIVE']:.3f} "
      f"(bank {v['SQ_LDS_BANK_CONFLICT'] / v['SQ_LDS_IDX_ACTIVE']:.3f}, address {v['SQ_LDS_ADDR_CONFLICT'] / v['SQ_LDS_IDX_ACTIVE']:.3f}), "
      f"LDS-array cycles per LDS instruction {v['SQ_LDS_IDX_ACTIVE'] / v['SQ_INSTS_LDS']:.2f}")
print(f"waves parked {v['SQ_WAIT_ANY'] / v['SQ_WAVE_CYCLES']:.3f}, issue stalls {v['SQ_WAIT_INST_ANY'] / v['SQ_WAVE_CYCLES']:.3f} "
      f"(LDS {v['SQ_WAIT_INST_LDS'] / v['SQ_WAVE_CYCLES']:.3f}); VALU instructions per LDS instruction {v['SQ_INSTS_VALU'] / v['SQ_INSTS_LDS']:.1f}")
