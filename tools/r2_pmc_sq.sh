#!/bin/bash
# SQ-level counters of the boundary kernels of the headline pass (where do their wave cycles go?)
export TMPDIR=/tmp
S=/tmp/pmc_sq; rm -rf $S; mkdir -p $S gpurun_out
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" "TCC_HIT_sum TCC_MISS_sum" "SQ_IFETCH SQ_ACTIVE_INST_VALU" ; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --output-format csv -d $S/p$i -- python3 tools/prof_driver.py 512 128 2 > $S/p$i.out 2>&1 || echo "pass $i ($C) failed: $(tail -2 $S/p$i.out)"
  echo pass $i done >> gpurun_out/pmc_sq_progress.txt
done
python3 - $S <<'PY'
import csv, glob, sys, collections
scr = sys.argv[1]
agg, nd = collections.defaultdict(float), collections.defaultdict(set)
for p in glob.glob(scr + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(p)):
        n = row["Kernel_Name"]
        if "rocclr" in n: continue
        key = (n[:44], row["Counter_Name"])
        agg[key] += float(row["Counter_Value"]); nd[key].add(row["Dispatch_Id"])
names = sorted({k[0] for k in agg})
ctrs = sorted({k[1] for k in agg})
with open("gpurun_out/pmc_sq.txt", "w") as o:
    for n in names:
        o.write(n + "\n")
        for c in ctrs:
            if (n, c) in agg: o.write("    %-22s %16.0f per launch\n" % (c, agg[(n, c)] / len(nd[(n, c)])))
print(open("gpurun_out/pmc_sq.txt").read())
PY
