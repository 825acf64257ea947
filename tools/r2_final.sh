#!/bin/bash
# final records of the round: full GPU suite, default bench line, headline under the profiler (kernel stats), rehearsal of N = 2
O=gpurun_out/final; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/gpu_suite.log 2>&1; tail -3 $O/gpu_suite.log
bash tools/r2_pmc.sh r02g > $O/pmc.log 2>&1; echo pmc done > $O/progress.txt
python bench.py > $O/default_bench.json 2> $O/default_bench.err; echo bench done >> $O/progress.txt
python bench.py --steps 3 --warmup 1 --no-cpu --base 256 --nlev 4 --box 64 --ncomp 55 > $O/c5shape.json 2> $O/c5shape.err
python bench.py --steps 20 --warmup 5 --no-cpu --sim-of 8 > $O/sim8.json 2> $O/sim8.err
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/final/*.json")) + ["gpurun_out/r02g_bench.json"]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print("%-28s %9.3f ms/step %9.1f Mcells/s  roofline %.3f step %.3f traffic %s" % (f.split("/")[-1], d["ms_per_step"], d["value"], d["roofline"]["frac"], d["step_frac_of_hbm_roofline"], d["roofline"]["traffic"]), d.get("breakdown_ms_per_step"), d.get("cpu_baseline", {}).get("value"))
    except Exception as e:
        print(f, "unreadable", e)
PY
cat gpurun_out/r02g_kstats.txt
