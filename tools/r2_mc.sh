#!/bin/bash
# marching-cubes level pass: [parity tests,] timing of both forms of the cell kernel, kernel stats and HBM counters
#   usage (GPU box): [TESTS=1] [KSEGS="16 48"] [PMC=1] tools/r2_mc.sh
O=gpurun_out/mc
mkdir -p $O
export TMPDIR=/tmp
if [ -n "$TESTS" ]; then
  python -m pytest tests/test_gpu_filter_mc.py tests/test_gpu_random.py -m gpu -x -q > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
  tail -3 $O/tests.log
fi
python tools/kernel_bench.py 512 128 mconly > $O/kb_new.json 2> $O/kb_new.err || exit 1
PA_MC_CELLS=tiles python tools/kernel_bench.py 512 128 mconly > $O/kb_old.json 2> $O/kb_old.err || exit 1
for k in ${KSEGS:-}; do PA_MC_KSEG=$k python tools/kernel_bench.py 512 128 mconly > $O/kb_kseg$k.json 2>/dev/null; done
rm -rf /tmp/mcprof; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mcprof/trace -- python3 tools/kernel_bench.py 512 128 mconly > $O/prof.log 2>&1
echo trace done > $O/progress.txt
if [ -n "$PMC" ]; then
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 400 rocprofv3 --pmc $C --output-format csv -d /tmp/mcprof/pmc_$C -- python3 tools/kernel_bench.py 512 128 mconly > /tmp/mcprof/pmc_$C.out 2>&1 || echo "pmc pass $C failed"
    echo pmc $C done >> $O/progress.txt
  done
fi
python3 - <<'PY'
import json, glob, csv, collections
for f in sorted(glob.glob("gpurun_out/mc/kb_*.json")):
    d = json.load(open(f))["kernels"]
    for k, v in d.items():
        if k.startswith("pa_mc_level"): print(f.split("/")[-1], k[:22], "%.3f ms  %.1f Gcells/s" % (v["ms"], v["Mcells_s"] / 1e3))
with open("gpurun_out/mc/kstats.txt", "w") as o:
    for p in glob.glob("/tmp/mcprof/trace/**/*kernel_stats.csv", recursive=True):
        for row in csv.DictReader(open(p)):
            n = row.get("Name", "")
            if "mcl" in n or "iso_mask" in n or "k_mc_" in n:
                o.write("%-70s calls %5s avg_ns %10s\n" % (n[:70], row.get("Calls"), row.get("AverageNs")))
    agg, nd = collections.defaultdict(float), collections.defaultdict(set)
    for p in glob.glob("/tmp/mcprof/pmc_*/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(p)):
            if "mcl" in row["Kernel_Name"]:
                key = (row["Kernel_Name"][:40], row["Counter_Name"])
                agg[key] += float(row["Counter_Value"]); nd[key].add(row["Dispatch_Id"])
    for key in sorted(agg):
        o.write("%-40s %-10s per launch %12.0f KiB (%d launches)%s\n" % (key[0], key[1], agg[key] / len(nd[key]), len(nd[key]), "  [x2 on gfx950]" if key[1] == "FETCH_SIZE" else ""))
print(open("gpurun_out/mc/kstats.txt").read())
PY
