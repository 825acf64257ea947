#!/bin/bash
# Profile the fused grad->curvature path on the GPU box: kernel trace + stats, then PMC passes in
# separate runs (as the MI355X guide prescribes).  Every run is bounded by `timeout`.
# Usage: tools/prof.sh <tag> [base] [box]   -> gpurun_out/prof_<tag>/summary.txt (small files only)
set -u
TAG=$1; BASE=${2:-256}; BOX=${3:-128}
OUT=$PWD/gpurun_out/prof_$TAG
SCR=/tmp/prof_$TAG
rm -rf "$SCR"; mkdir -p "$OUT" "$SCR"
export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$SCR/trace" -- python3 tools/prof_driver.py $BASE $BOX 3 > "$OUT/driver.txt" 2> "$SCR/trace.err" || echo "trace run failed/timeout" >> "$OUT/driver.txt"
i=0
for P in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
         "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM" \
         "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $P --output-format csv -d "$SCR/pmc_$i" -- python3 tools/prof_driver.py $BASE $BOX 1 > /dev/null 2> "$SCR/pmc_$i.err" || echo "pmc pass $i ($P) failed/timeout" >> "$OUT/driver.txt"
done
python3 - "$SCR" "$OUT" <<'PY'
import csv, glob, sys, collections, os
scr, out = sys.argv[1], sys.argv[2]
with open(os.path.join(out, "summary.txt"), "w") as f:
    for p in glob.glob(scr + "/trace/**/*kernel_stats.csv", recursive=True):
        f.write("== kernel stats (%s)\n" % os.path.basename(p))
        for row in csv.DictReader(open(p)):
            f.write("%-64s calls %6s total_ns %14s avg_ns %12s pct %s\n" % (row.get("Name", "")[:64], row.get("Calls"), row.get("TotalDurationNs"), row.get("AverageNs"), row.get("Percentage")))
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    nd = collections.defaultdict(set)
    for p in glob.glob(scr + "/pmc_*/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(p)):
            k = row["Kernel_Name"][:64]
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
            nd[(k, row["Counter_Name"])].add(row["Dispatch_Id"])
    f.write("== PMC: sum over dispatches / number of dispatches (per kernel)\n")
    for k, cs in agg.items():
        f.write(k + "\n")
        for c, v in sorted(cs.items()):
            n = len(nd[(k, c)])
            f.write("    %-28s sum %.6g  dispatches %d  per-dispatch %.6g\n" % (c, v, n, v / max(n, 1)))
print(open(os.path.join(out, "summary.txt")).read())
PY
