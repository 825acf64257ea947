#!/usr/bin/env python3
"""Experiment: residency of the narrow sweep's workgroups from the per-workgroup {start, end, HW_ID, XCC_ID} records that
PA_SWEEP_DBG=<file> made an INSTRUMENTED build of the library dump (level 0 of the hierarchy; the instrumentation -- wall_clock64 and
HW_ID / XCC_ID of lane 0 at the start and end of a workgroup of gradcurv_march3n_body -- was removed after the measurement, see
profiles/r04_small_experiments.txt).  usage: wg_residency.py <file>"""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.int64).reshape(-1, 4)
a = a[a[:, 0] != 0]
t0, t1, hw, xcc = a[:, 0], a[:, 1], a[:, 2], a[:, 3] & 0xf
cu = (hw >> 8) & 0xf
sh = (hw >> 12) & 0x1
se = (hw >> 13) & 0x7
key = xcc * 1000 + se * 100 + sh * 20 + cu
dur = (t1 - t0) / 100.0  # wall_clock64: 100 MHz -> us
span = (t1.max() - t0.min()) / 100.0
print("workgroups %d, span %.1f us, duration per workgroup: median %.1f us (min %.1f, max %.1f)" % (len(a), span, np.median(dur), dur.min(), dur.max()))
print("distinct (xcc, se, sh, cu) keys: %d; xcc values %s; se values %s; cu values %s" % (len(np.unique(key)), np.unique(xcc), np.unique(se), np.unique(cu)))
# average number of workgroups resident on the chip, and per key
print("average workgroups resident on the chip: %.1f  (= sum of durations / span)" % (dur.sum() / span))
per = {}
for k in np.unique(key):
    m = key == k
    per[k] = dur[m].sum() / span
v = np.array(list(per.values()))
print("per key: mean %.2f  min %.2f  max %.2f workgroups resident" % (v.mean(), v.min(), v.max()))
# gaps between consecutive workgroups on one key (time the slot pair is below 2)
k0 = np.unique(key)[0]
m = key == k0
ev = sorted([(s, 1) for s in t0[m]] + [(e, -1) for e in t1[m]])
lvl, last, hist = 0, ev[0][0], {}
for t, d in ev:
    hist[lvl] = hist.get(lvl, 0) + (t - last) / 100.0
    lvl += d; last = t
print("key %d: time (us) with n workgroups resident:" % k0, {k: round(v, 1) for k, v in sorted(hist.items())})
