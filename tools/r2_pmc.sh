# Round-2 evidence for the bench line: the driver's bench command under rocprofv3 --kernel-trace --stats, then the same
# command under two separate --pmc passes (FETCH_SIZE, WRITE_SIZE: the guide's HBM section), summarised into
#   gpurun_out/<tag>_bench.json (the bench line), gpurun_out/<tag>_kstats.txt (kernel stats), gpurun_out/<tag>_traffic.json
# usage (on the GPU box): bash tools/r2_pmc.sh <tag> [bench args]
TAG=${1:-r02}; shift
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
S=/tmp/pmc_$TAG; rm -rf $S; mkdir -p $S
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $S/trace -- python3 bench.py --steps 20 --warmup 5 "$@" > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
echo "trace done" > gpurun_out/${TAG}_progress.txt
# counter passes on the torch-free driver (same hierarchy, same library calls; torch's data generation is ~10^4 tiny
# dispatches, each serialised under counter collection)
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --output-format csv -d $S/pmc_$C -- python3 tools/prof_driver.py 512 128 2 > $S/pmc_$C.out 2> $S/pmc_$C.err || echo "pmc pass $C failed" >> gpurun_out/${TAG}_bench.err
  echo "pmc $C done" >> gpurun_out/${TAG}_progress.txt
done
python3 - $S $TAG <<'PY'
import csv, glob, json, sys, collections
scr, tag = sys.argv[1], sys.argv[2]
with open(f"gpurun_out/{tag}_kstats.txt", "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 5 (kernels of the library; torch's data-generation kernels omitted)\n")
    sym, avg = None, None
    for p in glob.glob(scr + "/trace/**/*kernel_stats.csv", recursive=True):
        for row in csv.DictReader(open(p)):
            n = row.get("Name", "")
            if "at::native" in n or "rocclr" in n or "elementwise" in n:
                continue
            f.write("%-100s calls %6s total_ns %12s avg_ns %12s pct %s\n" % (n[:100], row.get("Calls"), row.get("TotalDurationNs"), row.get("AverageNs"), row.get("Percentage")))
            if "k_gradcurv_march3" in n:
                sym, avg, calls = n, float(row["AverageNs"]), int(row["Calls"])
agg, nd = collections.defaultdict(float), collections.defaultdict(set)
for p in glob.glob(scr + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(p)):
        if "k_gradcurv_march3" in row["Kernel_Name"]:
            agg[row["Counter_Name"]] += float(row["Counter_Value"])
            nd[row["Counter_Name"]].add(row["Dispatch_Id"])
line = json.loads(open(f"gpurun_out/{tag}_bench.json").read().strip().splitlines()[-1])
rec = {"command": "tools/r2_pmc.sh: rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 5 ; rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE -- python3 tools/prof_driver.py 512 128 2 (separate passes, the same hierarchy and library calls without torch)",
       "workload": line["config"]["workload"], "kernel": line["roofline"]["kernel"].split(" (")[0], "symbol": sym,
       "avg_launch_ns_trace": avg, "launches_in_trace": calls, "avg_launch_ms_bench_events": line["roofline"]["avg_launch_ms"]}
if "FETCH_SIZE" in agg and "WRITE_SIZE" in agg:
    fetch, write = agg["FETCH_SIZE"] / len(nd["FETCH_SIZE"]), agg["WRITE_SIZE"] / len(nd["WRITE_SIZE"])
    rec.update({"FETCH_SIZE_KiB_per_launch": fetch, "WRITE_SIZE_KiB_per_launch": write,
                "note": "gfx950 correction per MI355X_MICROARCH.md (HBM section): FETCH_SIZE counts 64 B per 128-B request -> doubled; WRITE_SIZE exact",
                "traffic_bytes_per_launch": int(2 * fetch * 1024 + write * 1024), "algorithmic_bytes_per_launch": int(line["roofline"]["cells_per_launch"] * 72)})
json.dump(rec, open(f"gpurun_out/{tag}_traffic.json", "w"), indent=1)
print(open(f"gpurun_out/{tag}_kstats.txt").read()); print(json.dumps(rec, indent=1)); print(json.dumps(line)[:1800])
PY
