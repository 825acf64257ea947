#!/bin/bash
# end-of-round evidence: headline (kernel stats + PMC traffic), then the other recorded configurations
O=gpurun_out/ev; mkdir -p $O
bash tools/r2_pmc.sh r02e > $O/pmc.log 2>&1
echo pmc done > $O/progress.txt
run() { tag=$1; shift; python bench.py "$@" > $O/$tag.json 2> $O/$tag.err; echo $tag done >> $O/progress.txt; }
run c2_1lev_512_10comp --steps 5 --warmup 2 --no-cpu --nlev 1 --ncomp 10
run c2_1lev_512_box32_1comp --steps 5 --warmup 2 --no-cpu --nlev 1 --box 32
run c5shape_4lev_256_box64_55comp --steps 3 --warmup 1 --no-cpu --base 256 --nlev 4 --box 64 --ncomp 55
run headline_box64 --steps 10 --warmup 3 --no-cpu --box 64
run headline_10comp --steps 3 --warmup 1 --no-cpu --ncomp 10
for n in 2 4 8; do run sim_rank0_of_$n --steps 20 --warmup 5 --no-cpu --sim-of $n; done
python bench.py > $O/default_bench.json 2> $O/default_bench.err
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/ev/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print("%-40s %9.3f ms/step %9.1f Mcells/s  roofline %.3f  step %.3f  %s" % (f.split("/")[-1], d["ms_per_step"], d["value"], d["roofline"]["frac"], d["step_frac_of_hbm_roofline"], d["roofline"]["kernel"][:48]), d["roofline"]["traffic"], d.get("cpu_baseline"))
    except Exception as e:
        print(f, "unreadable", e)
PY
