"""SQ counter passes of rocprofv3 (--pmc ... --kernel-trace, one directory per pass: <root>/pass1, pass2 ...) condensed per
kernel: the LAST `--last` dispatches of every kernel whose name matches the regex (the thermalised launches of the bench,
not the cold-lattice pre-roll), summed over the counter's instances, averaged over the dispatches; then the ratios the
roofline discussion uses.  Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles
summed over waves; SQ_LDS_* count LDS-array cycles.

    python scripts/sq_summary.py <root> <kernel regex> [--last N] [--waves-per-simd W --waves-per-cu C]"""
import argparse
import csv
import glob
import re
from collections import defaultdict

ap = argparse.ArgumentParser()
ap.add_argument("root")
ap.add_argument("regex")
ap.add_argument("--last", type=int, default=6)
args = ap.parse_args()
pat = re.compile(args.regex)
acc = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))   # kernel -> counter -> dispatch -> value
dur = defaultdict(dict)                                               # kernel -> dispatch -> ns
res = {}
for f in glob.glob(args.root + "/pass*/**/*counter_collection.csv", recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row.get("Kernel_Name", "")
            if not pat.search(name):
                continue
            acc[name][row["Counter_Name"]][(f, int(row["Dispatch_Id"]))] += float(row["Counter_Value"])
            res[name] = (row.get("VGPR_Count"), row.get("Accum_VGPR_Count"), row.get("SGPR_Count"), row.get("LDS_Block_Size"),
                         row.get("Scratch_Size"), row.get("Workgroup_Size"), row.get("Grid_Size"))
for f in glob.glob(args.root + "/pass*/**/*kernel_trace.csv", recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row.get("Kernel_Name", "")
            if pat.search(name):
                dur[name][(f, int(row["Dispatch_Id"]))] = float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
for name, ctrs in acc.items():
    print("==", name[:200])
    print("   VGPR, AGPR, SGPR, LDS B, scratch B, workgroup, grid:", res[name])
    v = {}
    for c, per in sorted(ctrs.items()):
        keys = sorted(per)
        byfile = defaultdict(list)
        for k in keys:
            byfile[k[0]].append(k)
        sel = [k for ks in byfile.values() for k in ks[-args.last:]]
        v[c] = sum(per[k] for k in sel) / max(len(sel), 1)
        print(f"   {c:26s} {v[c]:.6g}   (last {len(sel)} of {len(keys)} dispatches)")
    if dur.get(name):
        byfile = defaultdict(list)
        for k in sorted(dur[name]):
            byfile[k[0]].append(dur[name][k])
        d = [x for ks in byfile.values() for x in ks[-args.last:]]
        print(f"   duration under the counters: {sum(d) / len(d) / 1e6:.3f} ms per launch (same dispatches)")
        v["_ms"] = sum(d) / len(d) / 1e6
    g = v.get
    if g("SQ_WAVE_CYCLES"):
        w = g("SQ_WAVE_CYCLES")
        for c in ("SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS",
                  "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_FLAT"):
            if g(c) is not None:
                print(f"   {c} / SQ_WAVE_CYCLES = {100 * g(c) / w:.1f} % of a wave's resident time")
    if g("SQ_LDS_IDX_ACTIVE"):
        bc, ac = g("SQ_LDS_BANK_CONFLICT") or 0.0, g("SQ_LDS_ADDR_CONFLICT") or 0.0
        print(f"   LDS conflicts: (SQ_LDS_BANK_CONFLICT + SQ_LDS_ADDR_CONFLICT) / SQ_LDS_IDX_ACTIVE = ({bc:.4g} + {ac:.4g}) / "
              f"{g('SQ_LDS_IDX_ACTIVE'):.4g} = {100 * (bc + ac) / g('SQ_LDS_IDX_ACTIVE'):.1f} %")
    if g("SQ_INSTS_VALU") and g("SQ_ACTIVE_INST_VALU"):
        print(f"   quad-cycles per VALU instruction: {g('SQ_ACTIVE_INST_VALU') / g('SQ_INSTS_VALU'):.2f};"
              f" per LDS instruction: {(g('SQ_ACTIVE_INST_LDS') or 0) / max(g('SQ_INSTS_LDS') or 1, 1):.2f}")
    if g("GRBM_GUI_ACTIVE") and g("_ms"):
        print(f"   effective clock GRBM_GUI_ACTIVE / duration = {g('GRBM_GUI_ACTIVE') / g('_ms') / 1e6:.2f} GHz (if the counter is per device)")
