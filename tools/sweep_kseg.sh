#!/bin/bash
for k in 16 32 43 64 86 128; do for pr in 0 1; do
  PA_PAIR=$pr PA_KSEG=$k python bench.py --no-cpu --steps 8 --warmup 1 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('kseg $k pair $pr', round(d['roofline']['avg_launch_ms'],4), round(d['ms_per_step'],3))
"
done; done
