#!/bin/bash
# PMC comparison of the fused sweep with the no-arithmetic emulation of its access pattern
# (tools/bench/membench4): memory-side counters, one rocprofv3 --pmc pass per counter group.
# Usage: tools/prof_cmp.sh <tag>  -> gpurun_out/cmp_<tag>.txt
set -u
TAG=$1
OUT=$PWD/gpurun_out/cmp_$TAG.txt
SCR=/tmp/cmp_$TAG
rm -rf "$SCR"; mkdir -p "$SCR" "$PWD/gpurun_out"
export TMPDIR=/tmp
i=0
for P in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum" \
         "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum" \
         "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_TAG_STALL_sum" \
         "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum" \
         "TCC_HIT_sum TCC_MISS_sum TCC_BUBBLE_sum" \
         "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" \
         "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
         "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum" \
         "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
         "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_BUSY_avr" \
         "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $P --output-format csv -d "$SCR/a_$i" -- tools/bench/membench4 prof > /dev/null 2> "$SCR/a_$i.err" || echo "membench pass $i ($P) failed" >> "$OUT"
  timeout -k 10 300 rocprofv3 --pmc $P --output-format csv -d "$SCR/b_$i" -- python3 tools/prof_driver.py 512 128 1 > /dev/null 2> "$SCR/b_$i.err" || echo "driver pass $i ($P) failed" >> "$OUT"
  echo "pass $i done"
done
python3 - "$SCR" "$OUT" <<'PY'
import csv, glob, sys, collections
scr, out = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
nd = collections.defaultdict(set)
for p in glob.glob(scr + "/*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(p)):
        k = row["Kernel_Name"][:40]
        if "march" not in k: continue
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        nd[(k, row["Counter_Name"])].add(row["Dispatch_Id"])
with open(out, "a") as f:
    names = sorted({c for k in agg for c in agg[k]})
    ks = sorted(agg)
    f.write("%-40s" % "counter (per dispatch)" + "".join("%28s" % k[:26] for k in ks) + "\n")
    for c in names:
        f.write("%-40s" % c + "".join("%28.6g" % (agg[k][c] / max(len(nd[(k, c)]), 1)) for k in ks) + "\n")
print(open(out).read())
PY
