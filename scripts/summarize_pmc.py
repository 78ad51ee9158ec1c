"""Aggregates rocprofv3 --pmc counter_collection CSVs: per kernel name, mean counter value per dispatch."""
import csv
import glob
import re
import sys
from collections import defaultdict

root = sys.argv[1]
pat = re.compile(sys.argv[2] if len(sys.argv) > 2 else ".")
acc = defaultdict(lambda: defaultdict(list))
for f in (glob.glob(root + "/pass*/**/*counter_collection.csv", recursive=True) +
          glob.glob(root + "pass*/**/*counter_collection.csv", recursive=True)):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row.get("Kernel_Name", "")
            if not pat.search(name):
                continue
            short = name.split("(")[0][-60:]
            acc[short][row["Counter_Name"]].append((row.get("Dispatch_Id"), float(row["Counter_Value"])))
for k, ctrs in acc.items():
    print("==", k)
    for c, vals in sorted(ctrs.items()):
        # counter values may be split per XCD/SE instance: sum per dispatch, then average
        per = defaultdict(float)
        for d, v in vals:
            per[d] += v
        n = len(per)
        print(f"   {c:28s} {sum(per.values()) / max(n, 1):.6g}   (dispatches {n})")
