#!/bin/bash
# same-box calibration: membench3 (1-D stream + march), membench4 (sweep emulation), membench5 (matrix subset)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( echo "== membench3"; timeout 120 tools/bench/membench3 | head -21
  echo "== membench4"; timeout 120 tools/bench/membench4 | head -12
  echo "== membench5 matrix"; timeout 200 tools/bench/membench5 6 matrix ) > gpurun_out/r03_calib.txt 2>&1
tail -60 gpurun_out/r03_calib.txt
