"""Per-dispatch durations from a rocprofv3 kernel_trace csv: python scripts/kernel_durations.py <csv> <regex>"""
import csv, re, sys
pat = re.compile(sys.argv[2])
rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        if pat.search(r["Kernel_Name"]):
            rows.append((int(r["Start_Timestamp"]), r["Kernel_Name"].split("(")[0][-50:], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
rows.sort()
for t, n, d in rows:
    print(f"{n:52s} {d:9.3f} ms")
