"""HBM traffic per launch of the step's main kernels from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in separate
runs, counters only -- no tracing), and the SQ counters of the two particle kernels (two more passes: VALU and LDS-array
busy fractions, LDS bank / address conflicts), of the PRODUCTION library with no variant switch, written to
bench.PMC_TRAFFIC_FILE with a fingerprint of the kernel sources; bench.py shows the numbers only while the fingerprint
matches.  Run on the GPU box from the repo root:
    python scripts/pmc_traffic.py [outdir]"""
import csv
import glob
import json
import os
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import PMC_TRAFFIC_FILE, kernel_sources_sha

out = os.path.abspath(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc_traffic")
os.makedirs(out, exist_ok=True)
KERNELS = {   # phase -> substring of the kernel name
    "CurrentDeposition": "deposit_tile_rows_kernel<3",
    "GatherAndPush": "gather_push_tile_kernel<3, 1, 0, true",
    "EvolveB": "evolve_b_kernel<wxa::StencilCfg<1, 1, 3, 1",
    "EvolveE": "evolve_e_kernel<wxa::StencilCfg<1, 1, 3, 1",
}
SORT_INTERVAL = 2   # bench.py's default: of the last six pushes three are the cycle's special push (COUNT | SCATTER) and three plain (push_sort.hpp)
bench_args = ["--steps", "6", "--warmup", "1", "--preroll", "40", "--sort-interval", str(SORT_INTERVAL), "--no-cpu-baseline",
              "--no-phase-pass", "--no-sanity"]
res = {k: {} for k in KERNELS}
names = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    d = os.path.join(out, ctr)
    cmd = ["rocprofv3", "--pmc", ctr, "--output-format", "csv", "-d", d, "-o", "pmc", "--", sys.executable,
           os.path.join(ROOT, "bench.py")] + bench_args
    env = dict(os.environ, TMPDIR="/tmp")
    r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True)
    open(os.path.join(out, ctr + ".log"), "w").write(r.stdout[-4000:] + "\n" + r.stderr[-4000:])
    per = defaultdict(lambda: defaultdict(float))   # kernel name -> dispatch id -> sum over counter instances
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") == ctr:
                    per[row["Kernel_Name"]][int(row["Dispatch_Id"])] += float(row["Counter_Value"])
    for phase, sub in KERNELS.items():
        for name, disp in per.items():
            if sub in name:
                last = [disp[i] for i in sorted(disp)[-6:]]   # the thermalised, timed steps
                res[phase][ctr] = sum(last) / len(last)
                names[phase] = name.split("(")[0][:100]
# ---- SQ passes of the two particle kernels (what the roofline's `limiter` quotes) ----
SQ_PASSES = [
    "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES",
    "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU GRBM_GUI_ACTIVE",
]
sq = {k: {} for k in ("CurrentDeposition", "GatherAndPush")}
for ipass, ctrs in enumerate(SQ_PASSES):
    d = os.path.join(out, f"sq{ipass + 1}")
    cmd = ["rocprofv3", "--pmc"] + ctrs.split() + ["--kernel-include-regex", "deposit_tile_rows|gather_push_tile", "--output-format", "csv",
                                                   "-d", d, "-o", "pmc", "--", sys.executable, os.path.join(ROOT, "bench.py")] + bench_args
    r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True)
    open(os.path.join(out, f"sq{ipass + 1}.log"), "w").write(r.stdout[-4000:] + "\n" + r.stderr[-4000:])
    per = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))   # kernel -> counter -> dispatch -> sum over instances
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                per[row["Kernel_Name"]][row["Counter_Name"]][int(row["Dispatch_Id"])] += float(row["Counter_Value"])
    for phase in sq:
        for name, cs in per.items():
            if KERNELS[phase] in name:
                for c, disp in cs.items():
                    last = [disp[i] for i in sorted(disp)[-6:]]
                    sq[phase][c] = sum(last) / len(last)
                sq[phase]["kernel"] = name.split("(")[0][:160]
N_CU, N_SIMD, N_SE = 256, 1024, 32
for phase, v in sq.items():
    if "SQ_BUSY_CYCLES" not in v:
        continue
    cyc = v["SQ_BUSY_CYCLES"] / N_SE   # cycles of the launch (every shader engine busy from start to end)
    v["derived"] = {
        "cycles_per_launch": cyc,
        "valu_busy_frac": 4.0 * v["SQ_ACTIVE_INST_VALU"] / N_SIMD / cyc,          # quad-cycles of VALU issue, summed over waves
        "lds_array_busy_frac": v["SQ_LDS_IDX_ACTIVE"] / N_CU / cyc,                # LDS-array cycles, summed over CUs
        "lds_conflict_frac": (v["SQ_LDS_BANK_CONFLICT"] + v["SQ_LDS_ADDR_CONFLICT"]) / v["SQ_LDS_IDX_ACTIVE"],
        "lds_array_cycles_per_lds_instruction": v["SQ_LDS_IDX_ACTIVE"] / v["SQ_INSTS_LDS"],
    }
    if "SQ_WAVE_CYCLES" in v and "SQ_WAIT_ANY" in v:
        v["derived"]["wave_waiting_frac"] = v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"]          # s_waitcnt / barrier
        v["derived"]["wave_issue_stall_frac"] = v["SQ_WAIT_INST_ANY"] / v["SQ_WAVE_CYCLES"]
rec = {
    "sq_source": "rocprofv3 --pmc (two passes: " + " | ".join(SQ_PASSES) + ") of the same command, production library, mean of the last "
                 "6 dispatches; SQ_ACTIVE_INST_* / SQ_WAIT_* / SQ_WAVE_CYCLES are quad-cycles summed over waves, SQ_LDS_* LDS-array "
                 "cycles, SQ_BUSY_CYCLES cycles summed over the 32 shader engines (MI355X_MICROARCH.md)",
    "sq": sq,
    "source": "scripts/pmc_traffic.py: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate runs) of bench.py " + " ".join(bench_args)
              + "; mean of the last 6 dispatches per kernel, summed over the counter instances (XCDs)",
    "sources_sha16": kernel_sources_sha(),
    "workload": {"ncell": 256, "ppc": 2, "order": 3, "deposition": "esirkepov", "pusher": "boris", "filter": True,
                 "sort_interval": SORT_INTERVAL},
    "correction": "HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: on gfx950 FETCH_SIZE counts 128-byte requests as 64 "
                  "bytes (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is taken as reported",
    "KiB_per_dispatch": {k: {"kernel": names.get(k, "?"), **v} for k, v in res.items() if len(v) == 2},
}
json.dump(rec, open(os.path.join(out, os.path.basename(PMC_TRAFFIC_FILE)), "w"), indent=1)
print(json.dumps(rec, indent=1))
