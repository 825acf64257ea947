#!/bin/bash
# kernel stats of the headline pass on the torch-free driver: average time of every library kernel
export TMPDIR=/tmp
S=/tmp/kstat; rm -rf $S; mkdir -p $S gpurun_out
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $S/trace -- python3 tools/prof_driver.py ${1:-512} ${2:-128} 6 > $S/trace.out 2>&1 || { tail -5 $S/trace.out; exit 1; }
tail -1 $S/trace.out
python3 - $S <<'PY'
import csv, glob, sys
for p in glob.glob(sys.argv[1] + "/trace/**/*kernel_stats.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(p)) if "rocclr" not in r["Name"]]
    tot = sum(float(r["AverageNs"]) for r in rows if "build_sfcode" not in r["Name"])
    for r in rows: print("%-64s calls %4s avg %9.1f us" % (r["Name"][:64], r["Calls"], float(r["AverageNs"]) / 1e3))
    print("sum of the per-pass kernels: %.1f us" % (tot / 1e3))
PY
