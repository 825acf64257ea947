#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace directory: per kernel calls / avg us / total ms (library kernels only), and the
GPU-busy timeline per pass when --passes N is given.   usage: ktrace_summary.py <dir> [skip_first_calls_fraction]"""
import csv, glob, sys, collections
d = sys.argv[1]
rows = []
for p in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        n = r["Kernel_Name"]
        if "at::native" in n or "rocclr" in n or "elementwise" in n:
            continue
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n))
rows.sort()
agg = collections.OrderedDict()
for s, e, n in rows:
    a = agg.setdefault(n, [0, 0])
    a[0] += 1; a[1] += e - s
tot = sum(a[1] for a in agg.values())
print("# %d dispatches, %.3f ms of kernel time" % (len(rows), tot / 1e6))
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-110s calls %5d avg_us %9.1f total_ms %9.3f pct %5.1f" % (n[:110], c, t / c / 1e3, t / 1e6, 100.0 * t / tot))
