#!/usr/bin/env python3
"""Timeline of the last call in a rocprofv3 --kernel-trace csv: every kernel after the last long gap, with its start
relative to the first, its duration and the idle time in front of it.
usage: python3 tools/trace_timeline.py <kernel_trace.csv> [gap_us_that_separates_calls=200]"""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
gap = float(sys.argv[2]) if len(sys.argv) > 2 else 200.0
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
cut = 0
for i in range(1, len(rows)):
    if (int(rows[i]["Start_Timestamp"]) - int(rows[i-1]["End_Timestamp"])) / 1e3 > gap:
        cut = i
rows = rows[cut:]
t0 = int(rows[0]["Start_Timestamp"])
prev = t0
busy = 0.0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f us  +%6.1f idle  %8.1f us  %s" % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, r["Kernel_Name"][:90]))
    busy += (e - s) / 1e3
    prev = e
print("total %.1f us, kernels %.1f us, %d launches" % ((prev - t0) / 1e3, busy, len(rows)))
