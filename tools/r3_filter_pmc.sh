#!/bin/bash
# HBM traffic of the filter kernels (FETCH_SIZE x 2 + WRITE_SIZE per the guide), separate --pmc passes
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
S=/tmp/fpmc; rm -rf $S; mkdir -p $S
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --output-format csv -d $S/$C -- python3 tools/kernel_bench.py 512 ${1:-128} filteronly > $S/$C.out 2>&1 || echo "pass $C failed"
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $S/trace -- python3 tools/kernel_bench.py 512 ${1:-128} filteronly > $S/trace.out 2>&1
python3 - $S <<'PY' | tee gpurun_out/r03_filter_traffic_${1:-128}.txt
import csv, glob, sys, collections
scr = sys.argv[1]
agg, nd = collections.defaultdict(float), collections.defaultdict(set)
for p in glob.glob(scr + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        n = r["Kernel_Name"]
        if "filter" not in n: continue
        k = (n[:60], r["Counter_Name"]); agg[k] += float(r["Counter_Value"]); nd[k].add(r["Dispatch_Id"])
dur = {}
for p in glob.glob(scr + "/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "filter" in r["Name"]: dur[r["Name"][:60]] = float(r["AverageNs"])
print("# rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) + --kernel-trace --stats -- python3 tools/kernel_bench.py 512 <box> filteronly")
print("# traffic = 2 x FETCH_SIZE (gfx950: 64 B counted per 128-B request) + WRITE_SIZE per launch; algorithmic = 2.147 GB (16 B x 512^3)")
for n in sorted({k[0] for k in agg}):
    f = agg[(n, "FETCH_SIZE")] / max(1, len(nd[(n, "FETCH_SIZE")])) * 1024 * 2 / 1e9
    w = agg[(n, "WRITE_SIZE")] / max(1, len(nd[(n, "WRITE_SIZE")])) * 1024 / 1e9
    print("%-62s avg %8.1f us  fetch x2 %6.3f GB  write %6.3f GB  total %6.3f GB = %.2f x algorithmic" % (n, dur.get(n, 0) / 1e3, f, w, f + w, (f + w) / 2.147))
PY
