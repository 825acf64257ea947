#!/bin/bash
# Round 3: the store-ceiling matrix (tools/bench/membench5.hip): timings, then memory-side counters per cell
# (one dispatch per cell under rocprofv3 --pmc; separate passes per counter group).  Run on the GPU box.
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B=tools/bench/membench5
timeout 300 $B 6 all > gpurun_out/r03_membench5.txt 2>&1 || { echo "membench5 failed"; tail -5 gpurun_out/r03_membench5.txt; exit 1; }
S=/tmp/mb5; rm -rf $S; mkdir -p $S
i=0
for C in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum" "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" "TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE" "TCC_EA0_WRREQ_IO_CREDIT_STALL_sum TCC_EA0_WRREQ_GMI_CREDIT_STALL_sum TCC_BUSY_avr"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --output-format csv -d $S/p$i -- $B 1 matrix > $S/p$i.out 2>&1 || echo "pass $i ($C) failed: $(tail -2 $S/p$i.out)" >> gpurun_out/r03_membench5_pmc.err
  echo "pass $i done" >> gpurun_out/r03_membench5_progress.txt
done
python3 - $S <<'PY'
import csv, glob, sys, collections
scr = sys.argv[1]
cells = [l.split("|")[0].strip() for l in open(scr + "/p1.out") if l.startswith("S ")]
tab = collections.defaultdict(dict)
for p in sorted(glob.glob(scr + "/p*/**/*counter_collection.csv", recursive=True)):
    rows = [r for r in csv.DictReader(open(p)) if "k_mix" in r["Kernel_Name"]]
    ids = sorted({int(r["Dispatch_Id"]) for r in rows})
    pos = {d: n for n, d in enumerate(ids)}
    for r in rows:
        tab[pos[int(r["Dispatch_Id"])]][r["Counter_Name"]] = float(r["Counter_Value"])
ctrs = sorted({c for v in tab.values() for c in v})
with open("gpurun_out/r03_membench5_pmc.txt", "w") as o:
    o.write("# rocprofv3 --pmc <group> -- tools/bench/membench5 1 matrix : one dispatch per cell, counters per dispatch\n")
    for n in sorted(tab):
        o.write((cells[n] if n < len(cells) else "cell %d" % n) + "\n")
        for c in ctrs:
            if c in tab[n]: o.write("    %-42s %16.0f\n" % (c, tab[n][c]))
print(open("gpurun_out/r03_membench5_pmc.txt").read()[:6000])
PY
cat gpurun_out/r03_membench5.txt
