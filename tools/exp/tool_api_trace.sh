#!/bin/bash
# experiment: where the first-call milliseconds of a tool's compute phase go (HIP API + kernel time of one run)
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
export TMPDIR=/tmp
D=/tmp/e2e_plt; rm -rf $D; mkdir -p $D
python3 - <<PY
import os, sys
sys.path.insert(0, ".")
from peleanalysis_amd.hierarchy import MultiFab, field_flame, fill_analytic, nested_hierarchy
from peleanalysis_amd.plotfile import write_plotfile
H = nested_hierarchy(256, 3, 64, is_per=(1, 1, 0))
mfs = []
for lv in H.levels:
    s = MultiFab(lv, 3, 0, fill=0.0)
    for c in range(3):
        fill_analytic(s, c, (lambda x, y, z, c=c: field_flame(x, y, z, c)))
    mfs.append(s)
write_plotfile("$D/plt00000", H, mfs, ["temp", "x_velocity", "density"], time=0.0, level_steps=[0, 0, 0])
PY
cd $D
for T in "$@"; do
  case $T in
    grad3d.ex) A="gradVar=temp is_per=1 1 0";;
    curvature3d.ex) A="progressName=temp is_per=1 1 0";;
    *) A="is_per=1 1 0";;
  esac
  PA_TOOL_EXIT=normal timeout 300 rocprofv3 --hip-trace --kernel-trace --stats --output-format csv -d /tmp/api_$T -- $GRAFT_REPO_ROOT/tools/bin/$T infile=$D/plt00000 bench_json=1 $A > /tmp/api_$T.out 2>&1
  echo "== $T"; grep '"tool"' /tmp/api_$T.out | tail -1
  find /tmp/api_$T -name "*.csv" | head; python3 - /tmp/api_$T <<'PY'
import csv, glob, sys
for pat, title in (("*hip_api_stats.csv", "HIP API"), ("*hip_stats.csv", "HIP API"), ("*kernel_stats.csv", "kernels")):
    for p in glob.glob(sys.argv[1] + "/**/" + pat, recursive=True):
        rows = list(csv.DictReader(open(p)))
        rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
        print("--", title)
        for r in rows[:14]:
            print("  %-70s calls %6s total_ms %9.3f" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e6))
PY
done
