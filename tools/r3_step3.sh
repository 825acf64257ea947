#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python3 bench.py --no-cpu --steps 20 --warmup 5 > gpurun_out/r03_s3_bench.json 2> gpurun_out/r03_s3_bench.err; cut -c1-1500 gpurun_out/r03_s3_bench.json
python3 bench.py --no-cpu --steps 20 --warmup 5 > gpurun_out/r03_s3_bench2.json 2>> gpurun_out/r03_s3_bench.err; cut -c1-300 gpurun_out/r03_s3_bench2.json
timeout 300 python tools/kernel_bench.py 512 128 filteronly > gpurun_out/r03_s3_filter128.json 2> gpurun_out/r03_s3_filter128.err; cat gpurun_out/r03_s3_filter128.json
timeout 1100 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r03_s3_tests.txt; cat gpurun_out/r03_s3_tests.txt
