#!/usr/bin/env python3
"""usage: timeline.py <trace dir> [launches of the marker kernel per step = 3] [marker kernel = k_gradcurv_march3]
Timeline of one bench step from a rocprofv3 --kernel-trace csv: per kernel start (us from the step's first
kernel), duration, stream/queue; and sum-of-durations against the busy union (how much ran concurrently)."""
import csv, sys, glob
path = sys.argv[1]
f = glob.glob(path + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last step: find the last 3 k_gradcurv_march3 launches and the window around them
pat = sys.argv[3] if len(sys.argv) > 3 else 'k_gradcurv_march3'  # the kernel that ends a step, launched nlev times per step
idx = [i for i, r in enumerate(rows) if pat in r['Kernel_Name']]
nlev = int(sys.argv[2]) if len(sys.argv) > 2 else 3
first_sweep = idx[-nlev]
prev_sweep = idx[-nlev - 1]
# the step starts after the previous step's last kernel: approximate by the kernel after prev step's final curv
lo = prev_sweep + 1
while lo < first_sweep and 'faces' in rows[lo]['Kernel_Name']: lo += 1
hi = idx[-1] + 1
while hi < len(rows) and 'faces' in rows[hi]['Kernel_Name']: hi += 1
sel = rows[lo:hi]
t0 = int(sel[0]['Start_Timestamp'])
tot = 0; ivs = []
for r in sel:
    s = int(r['Start_Timestamp']) - t0; e = int(r['End_Timestamp']) - t0
    tot += e - s; ivs.append((s, e))
    print(f"{s/1e3:9.1f} {(e-s)/1e3:8.1f}  q{r.get('Queue_Id','?'):>3} s{r.get('Stream_Id','?'):>3}  {r['Kernel_Name'][:60]}")
ivs.sort(); busy = 0; cs, ce = ivs[0]
for s, e in ivs[1:]:
    if s > ce: busy += ce - cs; cs, ce = s, e
    else: ce = max(ce, e)
busy += ce - cs
print(f"span {(max(e for _, e in ivs))/1e3:.1f} us  sum {tot/1e3:.1f} us  busy-union {busy/1e3:.1f} us")
