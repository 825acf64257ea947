#!/usr/bin/env python3
"""usage: ktimeline.py <rocprofv3 --kernel-trace dir> <marker kernel substring> [passes = 2]
The library's kernels of the last `passes` passes (a pass ends with the marker kernel): start (us from the first one), duration, queue."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "at::native" not in r["Kernel_Name"] and "elementwise" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if sys.argv[2] in r["Kernel_Name"]]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 2
lo = idx[-n - 1] + 1 if len(idx) > n else 0
t0 = int(rows[lo]["Start_Timestamp"])
for r in rows[lo:idx[-1] + 1]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%9.1f %8.1f q%s %s" % (s / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"][:100]))
