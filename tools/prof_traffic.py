#!/usr/bin/env python3
"""summary.txt of tools/prof.sh -> profiles/<tag>_traffic.json (HBM bytes per launch of the fused sweep,
corrected as MI355X_MICROARCH.md prescribes: FETCH_SIZE on gfx950 counts 64 B per 128-B request ->
doubled; WRITE_SIZE exact).  usage: prof_traffic.py <summary.txt> <out.json> [base box]"""
import json
import re
import sys

txt = open(sys.argv[1]).read()
base, box = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (512, 128)
m = re.search(r"^(void k_gradcurv_march3?<[^\n]*?)\s+calls\s+(\d+)\s+total_ns\s+(\d+)\s+avg_ns\s+([\d.]+)", txt, re.M)
name, calls, avg_ns = m.group(1).strip(), int(m.group(2)), float(m.group(4))
blk = txt[txt.index("== PMC"):]
blk = blk[blk.index(name[:40]):]
fetch = float(re.search(r"FETCH_SIZE\s+sum \S+\s+dispatches \d+\s+per-dispatch (\S+)", blk).group(1))
write = float(re.search(r"WRITE_SIZE\s+sum \S+\s+dispatches \d+\s+per-dispatch (\S+)", blk).group(1))
traffic = int(2 * fetch * 1024 + write * 1024)
json.dump({
    "command": f"tools/prof.sh <tag> {base} {box}   (rocprofv3 --kernel-trace --stats, then separate --pmc passes; python3 tools/prof_driver.py {base} {box})",
    "workload": f"fused grad->curvature, 3-level AMR, base {base}^3, {box}^3 boxes, 1 comp (same as bench.py default)",
    "kernel": name, "avg_launch_ns": avg_ns, "launches_in_trace": calls,
    "FETCH_SIZE_KiB_per_launch": fetch, "WRITE_SIZE_KiB_per_launch": write,
    "note": "gfx950 correction per MI355X_MICROARCH.md (HBM section): FETCH_SIZE counts 64 B per 128-B request -> doubled; WRITE_SIZE exact",
    "traffic_bytes_per_launch": traffic, "algorithmic_bytes_per_launch": base ** 3 * 72,
}, open(sys.argv[2], "w"), indent=1)
print(open(sys.argv[2]).read())
