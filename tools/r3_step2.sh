#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_filter_mc.py tests/test_golden.py -x -q -m gpu -k "filter" 2>&1 | tail -15 > gpurun_out/r03_s2_tests.txt
cat gpurun_out/r03_s2_tests.txt
timeout 300 python tools/kernel_bench.py 512 128 filteronly > gpurun_out/r03_s2_filter128.json 2> gpurun_out/r03_s2_filter128.err; cat gpurun_out/r03_s2_filter128.json; tail -3 gpurun_out/r03_s2_filter128.err
timeout 300 python tools/kernel_bench.py 512 32 filteronly > gpurun_out/r03_s2_filter32.json 2> gpurun_out/r03_s2_filter32.err; cat gpurun_out/r03_s2_filter32.json
timeout 600 python -m pytest tests/test_plotfile_tools.py -x -q -m gpu -k "sparse_levels or filter" 2>&1 | tail -30 > gpurun_out/r03_s2_tools.txt
cat gpurun_out/r03_s2_tools.txt
timeout 300 tools/bench/membench5 6 stride > gpurun_out/r03_membench5_stride.txt 2>&1; cat gpurun_out/r03_membench5_stride.txt
