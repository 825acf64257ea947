# kernel trace + stats of the bench command (rocprofv3), summary to gpurun_out/<tag>_kstats.txt
TAG=${1:-r2}; shift
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf /tmp/tr_$TAG
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_$TAG -- python3 bench.py --steps 10 --warmup 3 --no-cpu "$@" > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python3 - /tmp/tr_$TAG gpurun_out/${TAG}_kstats.txt <<'PY'
import csv, glob, sys, os
scr, out = sys.argv[1], sys.argv[2]
with open(out, "w") as f:
    for p in glob.glob(scr + "/**/*kernel_stats.csv", recursive=True):
        for row in csv.DictReader(open(p)):
            f.write("%-90s calls %6s total_ns %12s avg_ns %10s pct %s\n" % (row.get("Name", "")[:90], row.get("Calls"), row.get("TotalDurationNs"), row.get("AverageNs"), row.get("Percentage")))
print(open(out).read())
PY
