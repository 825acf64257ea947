#!/bin/bash
# A/B of one environment switch on the headline bench: tools/r2_ab.sh VAR  (VAR=0 vs default), 2 repetitions each
O=gpurun_out/ab; mkdir -p $O
V=$1
for rep in 1 2; do
  python bench.py --steps 20 --warmup 5 --no-cpu > $O/on$rep.json 2> $O/on$rep.err
  env $V=0 python bench.py --steps 20 --warmup 5 --no-cpu > $O/off$rep.json 2> $O/off$rep.err
done
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/ab/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print("%-12s %8.3f ms/step %9.1f Mcells/s" % (f.split("/")[-1], d["ms_per_step"], d["value"]), d["breakdown_ms_per_step"])
PY
