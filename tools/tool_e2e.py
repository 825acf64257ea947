#!/usr/bin/env python3
"""End-to-end wall time of the plotfile tools on a synthetic C3-shaped plotfile (3 levels, base N^3, flame field, 3
components).  usage: python tools/tool_e2e.py [base=256] [box=64] [options]
"options": a 5-component file (temp, 3 velocity components, density) and curvature3d with do_gaussCurv + do_strain + do_velnormal only.
"smooth": curvature3d with do_smooth=1, smoothing_time 1e-7 (dt / dx^2 = 0.42 on the finest level at base 512: plain BiCGStab) and 1e-5 (42: preconditioned)."""
import os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from peleanalysis_amd.hierarchy import MultiFab, field_flame, fill_analytic, nested_hierarchy
from peleanalysis_amd.plotfile import write_plotfile
base = int(sys.argv[1]) if len(sys.argv) > 1 else 256
box = int(sys.argv[2]) if len(sys.argv) > 2 else 64
options = len(sys.argv) > 3 and sys.argv[3] == "options"
smooth = len(sys.argv) > 3 and sys.argv[3] == "smooth"
names = ["temp", "x_velocity", "y_velocity", "z_velocity", "density"] if options else ["temp", "x_velocity", "density"]
H = nested_hierarchy(base, 3, box, is_per=(1, 1, 0))
mfs = []
for lv in H.levels:
    s = MultiFab(lv, len(names), 0, fill=0.0)
    for c in range(len(names)):
        fill_analytic(s, c, (lambda x, y, z, c=c: field_flame(x, y, z, c)))
    mfs.append(s)
d = tempfile.mkdtemp(dir=os.environ.get("TMPDIR", "/tmp"))
p = os.path.join(d, "plt00000")
write_plotfile(p, H, mfs, names, time=0.0, level_steps=[0, 0, 0])
print(f"plotfile: base {base}^3, 3 levels, {box}^3 boxes, {sum(l.ncells for l in H.levels)} cells x {len(names)} comps", flush=True)
bindir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin")
runs = (("grad3d.ex", ["gradVar=temp", "is_per=1 1 0"]), ("curvature3d.ex", ["progressName=temp", "is_per=1 1 0"]),
        ("filterPlt3d.ex", ["is_per=1 1 0"]), ("isosurface3d.ex", ["isoCompName=temp", "isoVal=1150", "comps=0 1 2"]))
if options:
    runs = (("curvature3d.ex", ["progressName=temp", "is_per=1 1 0", "do_gaussCurv=1", "do_strain=1", "do_velnormal=1"]),
            ("curvature3d.ex", ["progressName=temp", "is_per=1 1 0", "do_gaussCurv=1", "do_strain=1", "do_velnormal=1", "fused=0"]))
if smooth:
    runs = (("curvature3d.ex", ["progressName=temp", "is_per=1 1 0", "do_smooth=1", "smoothing_time=1e-7"]),
            ("curvature3d.ex", ["progressName=temp", "is_per=1 1 0", "do_smooth=1", "smoothing_time=1e-5"]))
for tool, args in runs:
    for rep in range(2):
        t0 = time.perf_counter()
        out = subprocess.run([os.path.join(bindir, tool), "infile=" + p, "bench_json=1"] + args, cwd=d, capture_output=True, text=True)
        dt = time.perf_counter() - t0
        assert out.returncode == 0, out.stderr[-500:]
    for f in os.listdir(d):  # the first run's output, renamed and kept by the second one as UtilCreateCleanDirectory does: tidy up outside the timing
        if ".old." in f:
            subprocess.run(["rm", "-rf", os.path.join(d, f)])
    js = [ln for ln in out.stdout.splitlines() if ln.startswith('{"tool"')]
    if smooth:
        print("   ", "; ".join(ln for ln in (out.stdout + out.stderr).splitlines() if "mooth" in ln or "iteration" in ln), flush=True)
    print(f"{tool:18s} wall {dt:.2f} s (second run)  {js[-1] if js else ''}", flush=True)
