"""Timeline of the last dispatches of a rocprofv3 kernel_trace csv: start, duration and the idle gap in front of each.

    python scripts/kernel_timeline.py <kernel_trace.csv> [last_n]

The sum of the gaps over a step is what a hipGraph of the step (or fewer, larger launches) could win back."""
import csv
import sys

rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
rows = rows[-n:]
t0 = rows[0][0]
busy = gaps = 0.0
prev_end = None
for s, e, name in rows:
    gap = 0.0 if prev_end is None else max(0.0, (s - prev_end) / 1e3)
    short = name.split("(")[0]
    short = short[:60] + (" ..." + short[-24:] if len(short) > 90 else short[60:])
    print(f"{(s - t0) / 1e6:10.3f} ms  +{(e - s) / 1e3:9.1f} us  gap {gap:7.1f} us  {short}")
    busy += (e - s) / 1e3
    gaps += gap
    prev_end = max(prev_end or e, e)
print(f"busy {busy / 1e3:.3f} ms, gaps {gaps / 1e3:.3f} ms over {len(rows)} dispatches ({(rows[-1][1] - t0) / 1e6:.3f} ms)")
