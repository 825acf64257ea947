#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_filter_mc.py tests/test_golden.py -x -q -m gpu -k "filter" 2>&1 | tail -3
for K in 64 128 32; do echo "PA_FILTER_SEP_KSEG=$K"; PA_FILTER_SEP_KSEG=$K timeout 300 python tools/kernel_bench.py 512 128 filteronly 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for k,v in d['kernels'].items():
    if 'sep' in k: print('  %-55s %.3f ms  frac %.3f'%(k,v['ms'],v['frac_hbm']))"; done | tee gpurun_out/r03_s5_filter.txt
