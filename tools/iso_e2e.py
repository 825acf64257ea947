#!/usr/bin/env python3
"""End-to-end timing of isosurface3d.ex on a synthetic C4-shaped plotfile (3 levels, base N^3, flame field,
temp + 2 mapped components, isoVal = 1150).  usage: python tools/iso_e2e.py [base=128] [box=32]"""
import os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from peleanalysis_amd.hierarchy import MultiFab, field_flame, fill_analytic, nested_hierarchy
from peleanalysis_amd.plotfile import read_mef, write_plotfile
base = int(sys.argv[1]) if len(sys.argv) > 1 else 128
box = int(sys.argv[2]) if len(sys.argv) > 2 else 32
H = nested_hierarchy(base, 3, box, is_per=(0, 0, 0))
t0 = time.perf_counter()
mfs = []
for lv in H.levels:
    s = MultiFab(lv, 3, 0, fill=0.0)
    for c in range(3):
        fill_analytic(s, c, (lambda x, y, z, c=c: field_flame(x, y, z, c)))
    mfs.append(s)
d = tempfile.mkdtemp(dir=os.environ.get("TMPDIR", "/tmp"))
p = os.path.join(d, "plt00000")
write_plotfile(p, H, mfs, ["temp", "x_velocity", "density"], time=0.0, level_steps=[0, 0, 0])
print(f"synthetic plotfile: base {base}^3, 3 levels, {box}^3 boxes, {sum(l.ncells for l in H.levels)} cells ({time.perf_counter() - t0:.1f} s to make)", flush=True)
exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", "isosurface3d.ex")
for rep in range(2):
    t0 = time.perf_counter()
    out = subprocess.run([exe, "infile=" + p, "isoCompName=temp", "isoVal=1150", "comps=0 1 2", "verbose=1"], capture_output=True, text=True)
    dt = time.perf_counter() - t0
    assert out.returncode == 0, out.stderr
    print(f"run {rep}: wall {dt:.2f} s")
    for ln in out.stdout.splitlines():
        if "time" in ln or "Nelts" in ln or "of which" in ln:
            print("   ", ln.strip())
