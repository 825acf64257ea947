#!/usr/bin/env python3
"""sum a rocprofv3 --pmc counter per kernel name: pmc_sum.py <dir> <counter> [name filter]"""
import csv, glob, sys, collections
d, ctr = sys.argv[1], sys.argv[2]
flt = sys.argv[3] if len(sys.argv) > 3 else ""
agg = collections.defaultdict(lambda: [0.0, set()])
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] == ctr and flt in row["Kernel_Name"]:
            key = (row["Kernel_Name"][:70], row.get("Grid_Size", ""))
            a = agg[key]
            a[0] += float(row["Counter_Value"]); a[1].add(row["Dispatch_Id"])
for (k, g), (v, ids) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print("%-72s grid %-10s dispatches %4d  %s per dispatch %.1f" % (k, g, len(ids), ctr, v / len(ids)))
