#!/bin/bash
# One PMC pass (instruction counts) : tools/prof_pmc.sh <tag> [base] [box] -> gpurun_out/prof_<tag>/pmc.txt
set -u
TAG=$1; BASE=${2:-256}; BOX=${3:-128}
OUT=$PWD/gpurun_out/prof_$TAG
SCR=/tmp/prof_$TAG
rm -rf "$SCR"; mkdir -p "$OUT" "$SCR"
export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d "$SCR/pmc" -- python3 tools/prof_driver.py $BASE $BOX 1 > /dev/null 2> "$SCR/pmc.err" || echo "pmc pass failed/timeout"
python3 - "$SCR" "$OUT" <<'PY'
import csv, glob, sys, collections, os
scr, out = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for p in glob.glob(scr + "/pmc/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(p)):
        agg[row["Kernel_Name"][:40]][row["Counter_Name"]] += float(row["Counter_Value"])
with open(os.path.join(out, "pmc.txt"), "w") as f:
    for k, cs in agg.items():
        w = max(cs.get("SQ_WAVES", 1), 1)
        f.write("%-40s waves %9.0f | per wave: VALU %7.0f SALU %7.0f SMEM %6.0f VMEM_RD %6.1f VMEM_WR %5.1f | ACTIVE_ANY(quad) %.3g BUSY %.3g\n" % (
            k, w, cs.get("SQ_INSTS_VALU", 0) / w, cs.get("SQ_INSTS_SALU", 0) / w, cs.get("SQ_INSTS_SMEM", 0) / w, cs.get("SQ_INSTS_VMEM_RD", 0) / w,
            cs.get("SQ_INSTS_VMEM_WR", 0) / w, cs.get("SQ_ACTIVE_INST_ANY", 0), cs.get("SQ_BUSY_CYCLES", 0)))
print(open(os.path.join(out, "pmc.txt")).read())
PY
