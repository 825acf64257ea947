#!/bin/bash
# Kernel trace + stats only (no PMC passes): tools/prof_trace.sh <tag> [base] [box] -> gpurun_out/prof_<tag>/summary.txt
set -u
TAG=$1; BASE=${2:-256}; BOX=${3:-128}
OUT=$PWD/gpurun_out/prof_$TAG
SCR=/tmp/prof_$TAG
rm -rf "$SCR"; mkdir -p "$OUT" "$SCR"
export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$SCR/trace" -- python3 tools/prof_driver.py $BASE $BOX 3 > "$OUT/driver.txt" 2> "$SCR/trace.err" || echo "trace run failed/timeout" >> "$OUT/driver.txt"
python3 - "$SCR" "$OUT" <<'PY'
import csv, glob, sys, os
scr, out = sys.argv[1], sys.argv[2]
with open(os.path.join(out, "summary.txt"), "w") as f:
    for p in glob.glob(scr + "/trace/**/*kernel_stats.csv", recursive=True):
        f.write("== kernel stats (%s)\n" % os.path.basename(p))
        for row in csv.DictReader(open(p)):
            f.write("%-64s calls %6s total_ns %14s avg_ns %12s pct %s\n" % (row.get("Name", "")[:64], row.get("Calls"), row.get("TotalDurationNs"), row.get("AverageNs"), row.get("Percentage")))
    import collections
    per = collections.defaultdict(list)
    for p in glob.glob(scr + "/trace/**/*kernel_trace.csv", recursive=True):
        for row in csv.DictReader(open(p)):
            per[row["Kernel_Name"][:48]].append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]) - int(row["Start_Timestamp"])))
    f.write("== per-dispatch durations (us, launch order)\n")
    for k, v in per.items():
        v.sort()
        f.write("%-48s %s\n" % (k, " ".join("%.0f" % (d / 1e3) for _, d in v[:24])))
print(open(os.path.join(out, "summary.txt")).read())
print(open(os.path.join(out, "driver.txt")).read())
PY
