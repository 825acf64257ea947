#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1000 python -m pytest tests/test_gpu_gradcurv.py tests/test_gpu_random.py tests/test_gpu_dist.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -8
for E in 1 0; do for B in 32; do PA_NARROW_CG=$E python3 bench.py --no-cpu --no-secondary --box $B | python3 -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('PA_NARROW_CG=$E box $B: value %.0f ms %.3f' % (l['value'], l['ms_per_step']), l['roofline']['kernel'][:50], 'launch %.3f x %d' % (l['roofline']['avg_launch_ms'], l['roofline']['launches']), l['breakdown_ms_per_step'])"; done; done | tee gpurun_out/r03_s8_narrow.txt
PA_NARROW_CG=1 python3 bench.py --no-cpu --no-secondary --box 32 --nlev 1 | python3 -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('1 level box 32: value %.0f ms %.3f' % (l['value'], l['ms_per_step']), l['roofline']['kernel'][:50], 'launch %.3f' % (l['roofline']['avg_launch_ms']), l['breakdown_ms_per_step'])" | tee -a gpurun_out/r03_s8_narrow.txt
