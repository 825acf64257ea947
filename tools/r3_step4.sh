#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_filter_mc.py tests/test_golden.py tests/test_gpu_random.py -x -q -m gpu -k "filter" 2>&1 | tail -5
for B in 128 64 32; do timeout 300 python tools/kernel_bench.py 512 $B filteronly 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['level'])
for k,v in d['kernels'].items(): print('  %-55s %.3f ms  frac %.3f'%(k,v['ms'],v['frac_hbm']))"; done | tee gpurun_out/r03_s4_filter.txt
S=/tmp/mb5f; rm -rf $S; mkdir -p $S
timeout 300 rocprofv3 --pmc TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum TCC_EA0_WRREQ_STALL_sum --output-format csv -d $S/p1 -- tools/bench/membench5 1 fronts > $S/p1.out 2>&1
python3 - $S <<'PY' | tee gpurun_out/r03_membench5_fronts_pmc.txt
import csv, glob, sys, collections
scr = sys.argv[1]
cells = [l.split("|")[0].strip() for l in open(scr + "/p1.out") if "mode 3" in l]
tab = collections.defaultdict(dict)
for p in sorted(glob.glob(scr + "/p*/**/*counter_collection.csv", recursive=True)):
    rows = [r for r in csv.DictReader(open(p)) if "k_mix" in r["Kernel_Name"]]
    ids = sorted({int(r["Dispatch_Id"]) for r in rows})
    pos = {d: n for n, d in enumerate(ids)}
    for r in rows:
        tab[pos[int(r["Dispatch_Id"])]][r["Counter_Name"]] = float(r["Counter_Value"])
print("# rocprofv3 --pmc ... -- tools/bench/membench5 1 fronts (one dispatch per cell)")
for n in sorted(tab):
    print((cells[n] if n < len(cells) else "cell %d" % n), " ".join("%s=%.3g" % (k.replace("TCC_","").replace("_sum",""), v) for k, v in sorted(tab[n].items())))
PY
