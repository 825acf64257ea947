#!/bin/bash
# component-stride pad (doubles past 16 MiB) against: the 13-row sweep emulation (membench4), the 16-row march (membench6 depth 1),
# the 1-D stream (membench3 first line)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( for P in 64 128 192 256 320 384 448 512 576 640 704 768 832 896 960 1024 1088 1152 1216 1280 1344 1408 1536 1600 1664 1792 1856 1920 1984; do
  a=$(MB_PAD=$P timeout 60 tools/bench/membench4 x | grep "work" | head -1 | sed 's/.*best \([0-9.]*\) ms.*/\1/')
  b=$(MB_QUICK=1 MB_PAD=$P timeout 60 tools/bench/membench6 | grep "depth 1 ty 16" | head -1 | sed 's/.*best \([0-9.]*\) ms.*/\1/')
  c=$(MB_QUICK=1 MB_PAD=$P timeout 60 tools/bench/membench3 | grep "stream 1-D cpl 1" | head -1 | sed 's/.*best \([0-9.]*\) ms.*/\1/')
  echo "pad $P doubles: sweep-emulation(13 rows) $a ms   march(16 rows) $b ms   1-D stream $c ms"
done ) > gpurun_out/r03_padsweep.txt 2>&1
cat gpurun_out/r03_padsweep.txt
