#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_gradcurv.py tests/test_gpu_random.py tests/test_gpu_dist.py -x -q -m gpu 2>&1 | tail -8
for T in -1 0.05 -1 0.05; do python3 bench.py --no-cpu --no-secondary --threshold $T | python3 -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('threshold $T: value %.0f ms %.3f' % (l['value'], l['ms_per_step']), l['roofline']['kernel'][:70], 'launch %.3f' % l['roofline']['avg_launch_ms'], l['breakdown_ms_per_step'])"; done | tee gpurun_out/r03_s7_clip.txt
PA_FUSED2_CLIP=0 python3 bench.py --no-cpu --no-secondary --threshold 0.05 | python3 -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('threshold 0.05 first pipeline: value %.0f ms %.3f' % (l['value'], l['ms_per_step']), l['roofline']['kernel'][:70], 'launch %.3f' % l['roofline']['avg_launch_ms'], l['breakdown_ms_per_step'])" | tee -a gpurun_out/r03_s7_clip.txt
