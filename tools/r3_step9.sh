#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1100 python -m pytest tests/test_plotfile_tools.py tests/test_gpu_stream.py tests/test_gpu_sdf.py -x -q -m gpu 2>&1 | tail -15
