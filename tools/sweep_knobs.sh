#!/bin/bash
# sweep of the launch knobs of the fused sweep (bench.py --no-cpu), one JSON line each
for m in 91 101 111 121 131; do for o in 0 2; do for k in 128 64; do
  PA_MTY=$m PA_ORDER=$o PA_KSEG=$k python bench.py --no-cpu --steps 6 --warmup 1 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('mty $m order $o kseg $k', round(d['roofline']['avg_launch_ms'],4), round(d['ms_per_step'],3))
"
done; done; done
