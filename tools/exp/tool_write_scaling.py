#!/usr/bin/env python3
"""Experiment: does the plotfile writer of the tools scale with its threads?  grad3d on a base-N^3 plotfile with PA_IO_THREADS = 16 / 4 / 1.
usage: tool_write_scaling.py [base=384] [box=128]"""
import json, os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from peleanalysis_amd.hierarchy import MultiFab, field_flame, fill_analytic, nested_hierarchy
from peleanalysis_amd.plotfile import write_plotfile
base = int(sys.argv[1]) if len(sys.argv) > 1 else 384
box = int(sys.argv[2]) if len(sys.argv) > 2 else 128
H = nested_hierarchy(base, 3, box, is_per=(1, 1, 0))
mfs = []
for lv in H.levels:
    s = MultiFab(lv, 1, 0, fill=0.0)
    fill_analytic(s, 0, lambda x, y, z: field_flame(x, y, z, 0))
    mfs.append(s)
d = tempfile.mkdtemp(dir=os.environ.get("TMPDIR", "/tmp"))
p = os.path.join(d, "plt00000")
write_plotfile(p, H, mfs, ["temp"], time=0.0, level_steps=[0, 0, 0])
print(f"plotfile: base {base}^3, 3 levels, {sum(l.ncells for l in H.levels)} cells x 1 comp", flush=True)
exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bin", "grad3d.ex")
for nt, nf in ((16, 0), (16, 0), (16, 0), (4, 0), (1, 0)):
    env = dict(os.environ, PA_IO_THREADS=str(nt))
    if nf:
        env["PA_PLT_NFILES"] = str(nf)
    out = subprocess.run([exe, "infile=" + p, "bench_json=1", "gradVar=temp", "is_per=1 1 0"], cwd=d, capture_output=True, text=True, env=env)
    js = [ln for ln in out.stdout.splitlines() if ln.startswith('{"tool"')]
    ph = json.loads(js[-1])["phases_s"] if js else {}
    print(f"PA_IO_THREADS={nt:2d} PA_PLT_NFILES={'default' if not nf else nf}: read {ph.get('read', 0):.2f} s  write {ph.get('write', 0):.2f} s  (5 comps out)", flush=True)
