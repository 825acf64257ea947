#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace csv: kernels of the LAST n dispatches matching a window (start marker kernel .. end),
with start offsets, durations and gaps -- what a pass looks like on the GPU's timeline.  usage: trace_summary.py <dir> <first-kernel-substr> [count]"""
import csv, glob, sys
d, first = sys.argv[1], sys.argv[2]
count = int(sys.argv[3]) if len(sys.argv) > 3 else 40
rows = []
for p in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(p)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
if not idx:
    sys.exit("no kernel matching " + first)
i0 = idx[-1]
# walk back to the last occurrence that begins a pass: use the last match whose previous match is > count kernels away, else the last one
t0 = int(rows[i0]["Start_Timestamp"])
prev_end = t0
for r in rows[i0:i0 + count]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f us  +%7.1f gap  %8.1f us  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, r["Kernel_Name"][:90]))
    prev_end = max(prev_end, e)
