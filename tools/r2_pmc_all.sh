#!/bin/bash
# FETCH_SIZE / WRITE_SIZE per launch of EVERY library kernel of the headline pass (torch-free driver), with kernel times
export TMPDIR=/tmp
S=/tmp/pmc_all; rm -rf $S; mkdir -p $S gpurun_out
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $S/trace -- python3 tools/prof_driver.py 512 128 4 > $S/trace.out 2>&1
echo trace done > gpurun_out/pmc_all_progress.txt
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --output-format csv -d $S/pmc_$C -- python3 tools/prof_driver.py 512 128 2 > $S/pmc_$C.out 2>&1 || echo "pmc pass $C failed"
  echo pmc $C done >> gpurun_out/pmc_all_progress.txt
done
python3 - $S <<'PY'
import csv, glob, sys, collections
scr = sys.argv[1]
avg = {}
for p in glob.glob(scr + "/trace/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(p)):
        avg[row["Name"]] = (float(row["AverageNs"]), int(row["Calls"]))
agg, nd = collections.defaultdict(float), collections.defaultdict(set)
for p in glob.glob(scr + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(p)):
        key = (row["Kernel_Name"], row["Counter_Name"])
        agg[key] += float(row["Counter_Value"]); nd[key].add(row["Dispatch_Id"])
with open("gpurun_out/pmc_all.txt", "w") as o:
    o.write("# rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) + --kernel-trace --stats -- python3 tools/prof_driver.py 512 128 (headline hierarchy)\n")
    o.write("# traffic = 2 x FETCH_SIZE (gfx950: 64 B counted per 128-B request) + WRITE_SIZE, per launch\n")
    for name in sorted(avg, key=lambda n: -avg[n][0] * avg[n][1]):
        f = agg.get((name, "FETCH_SIZE"), 0) / max(1, len(nd.get((name, "FETCH_SIZE"), [1])))
        w = agg.get((name, "WRITE_SIZE"), 0) / max(1, len(nd.get((name, "WRITE_SIZE"), [1])))
        t = avg[name][0]
        gb = (2 * f + w) * 1024 / 1e9
        o.write("%-62s %9.1f us  fetch x2 %8.3f GB  write %8.3f GB  -> %6.2f TB/s\n" % (name[:62], t / 1e3, 2 * f * 1024 / 1e9, w * 1024 / 1e9, gb / (t * 1e-9) / 1e3 if t else 0))
print(open("gpurun_out/pmc_all.txt").read())
PY
