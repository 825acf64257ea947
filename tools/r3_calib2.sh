#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( for P in 64 256 128; do
  echo "== membench3 MB_PAD=$P"; MB_PAD=$P timeout 120 tools/bench/membench3 | head -11
  echo "== membench4 MB_PAD=$P"; MB_PAD=$P timeout 120 tools/bench/membench4 | head -4
done ) > gpurun_out/r03_calib2.txt 2>&1
cat gpurun_out/r03_calib2.txt
