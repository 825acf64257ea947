#!/bin/bash
# Round 3: how much of the sweep's time is arithmetic / LDS / stores in the CURRENT step schedule (diagnostic builds, wrong results
# on purpose): PA_DBG bits 1 = no stores, 2 = sqrt/divide replaced by adds, 4 = no LDS neighbour reads.  First pipeline (non-CG kernel).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for D in 0 2 4 6 1 7 0; do
  if [ $D = 0 ]; then PA_FUSED2=0 python3 bench.py --no-cpu --steps 10 --warmup 3 > /tmp/d.json 2>/tmp/d.err; else PA_FUSED2=0 PA_DBG=$D python3 bench.py --no-cpu --steps 10 --warmup 3 > /tmp/d.json 2>/tmp/d.err; fi
  python3 -c "
import json,sys
l=json.loads(open('/tmp/d.json').read().strip().splitlines()[-1])
print('PA_DBG=$D sweep avg launch %.3f ms  step %.3f ms' % (l['roofline']['avg_launch_ms'], l['ms_per_step']))" | tee -a gpurun_out/r03_dbg.txt
done
