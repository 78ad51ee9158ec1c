"""HBM traffic per launch of the step's main kernels from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in separate
runs, counters only -- no tracing), written to profiles/round3/r3_pmc_traffic.json with a fingerprint of the kernel
sources; bench.py shows the numbers only while the fingerprint matches.  Run on the GPU box from the repo root:
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
bench_args = ["--steps", "6", "--warmup", "1", "--preroll", "40", "--no-cpu-baseline", "--no-phase-pass", "--no-sanity"]
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
rec = {
    "source": "scripts/pmc_traffic.py: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate runs) of bench.py " + " ".join(bench_args)
              + "; mean of the last 6 dispatches per kernel, summed over the counter instances (XCDs)",
    "sources_sha16": kernel_sources_sha(),
    "workload": {"ncell": 256, "ppc": 2, "order": 3, "deposition": "esirkepov", "pusher": "boris", "filter": True,
                 "sort_interval": 3},
    "correction": "HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: on gfx950 FETCH_SIZE counts 128-byte requests as 64 "
                  "bytes (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is taken as reported",
    "KiB_per_dispatch": {k: {"kernel": names.get(k, "?"), **v} for k, v in res.items() if len(v) == 2},
}
json.dump(rec, open(os.path.join(out, os.path.basename(PMC_TRAFFIC_FILE)), "w"), indent=1)
print(json.dumps(rec, indent=1))
