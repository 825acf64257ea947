#!/bin/bash
# level-batched sweeps: parity tests, then A/B of the batched launch on the headline, BASELINE config 5's shape and rank 0's share of 8
O=gpurun_out/batch
mkdir -p $O
python -m pytest tests/test_gpu_gradcurv.py tests/test_gpu_random.py tests/test_gpu_dist.py tests/test_golden.py -m gpu -x -q > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
run() { tag=$1; shift; "$@" > $O/$tag.json 2> $O/$tag.err; python3 - $O/$tag.json $tag <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-28s %8.3f ms/step %9.1f Mcells/s  sweep %s x %.4f ms  %s" % (sys.argv[2], d["ms_per_step"], d["value"], d["roofline"]["launches"], d["roofline"]["avg_launch_ms"], d["roofline"]["kernel"][:60]), d.get("breakdown_ms_per_step"))
PY
}
run head_batch python bench.py --steps 20 --warmup 5 --no-cpu
PA_SWEEP_BATCH=0 run head_nobatch python bench.py --steps 20 --warmup 5 --no-cpu
run c5_batch python bench.py --steps 3 --warmup 1 --no-cpu --base 256 --nlev 4 --box 64 --ncomp 55
PA_SWEEP_BATCH=0 run c5_nobatch python bench.py --steps 3 --warmup 1 --no-cpu --base 256 --nlev 4 --box 64 --ncomp 55
run sim8_batch python bench.py --steps 20 --warmup 5 --no-cpu --sim-of 8
PA_DIST_SWEEP_BATCH=0 run sim8_nobatch python bench.py --steps 20 --warmup 5 --no-cpu --sim-of 8
run sim4_batch python bench.py --steps 20 --warmup 5 --no-cpu --sim-of 4
PA_DIST_SWEEP_BATCH=0 run sim4_nobatch python bench.py --steps 20 --warmup 5 --no-cpu --sim-of 4
