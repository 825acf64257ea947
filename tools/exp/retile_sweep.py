"""retile_sweep.py CASE [MAX ...] -- the fused grad->curvature pass on one hierarchy under several internal tilings.
CASE: irregular | box32 | box64 | c5 (the bench's secondary hierarchies).  MAX: "file" (the file's boxes) or "mx,my,mz"
(pa_level_retile limits).  Every tiling is allocated, run once, timed (3 passes) and released in turn; prints boxes per
level, the share of cells in boxes <= 32 wide, ms per pass, the fraction of 8 TB/s, and per-kernel-family ms from the
library's HIP events (sweep / fix-up / FillBoundary / applyBC)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from peleanalysis_amd import capi
from peleanalysis_amd.hierarchy import Hierarchy, tagged_hierarchy, nested_hierarchy, field_flame, mf_layout, retile_level
import bench

case = sys.argv[1]
maxes = sys.argv[2:] or ["file", "128,128,128", "256,128,128"]
ncomp, nbatch = 1, 1
if case == "irregular":
    H = tagged_hierarchy(512, 3, lambda x, y, z: field_flame(x, y, z, 0), bf=16, max_box=128, base_box=128, frac=(0.08, 0.16), is_per=(1, 1, 0))
elif case == "box32":
    H = nested_hierarchy(512, 3, 32, is_per=(1, 1, 0))
elif case == "box64":
    H = nested_hierarchy(512, 3, 64, is_per=(1, 1, 0))
elif case == "box128":
    H = nested_hierarchy(512, 3, 128, is_per=(1, 1, 0))
elif case == "c5":
    H = nested_hierarchy(256, 4, 64, is_per=(1, 1, 0))
    ncomp = nbatch = 8
else:
    raise SystemExit("unknown case")
dev = torch.device("cuda:0")
stream = torch.cuda.Stream(device=dev)
ctx = capi.Context(0, stream.cuda_stream)
bc = capi.bc_from_flags((1, 1, 0))
P = capi.curv_params(prog_min=300.0, prog_max=2003.0 * (1.0 + 0.1 * ncomp) + 3.0 * ncomp, threshold=None, fused=True)
cells = sum(lv.ncells for lv in H.levels) * ncomp
for m in maxes:
    def auto(lv):  # 256^3 where the level is made of large blocks, else 128^3
        r = retile_level(lv, (256, 256, 256))
        d = r.boxes[:, 3:] - r.boxes[:, :3] + 1
        return r if d.min() >= 128 else retile_level(lv, (128, 128, 128))
    T = H if m == "file" else Hierarchy([auto(lv) if m == "auto" else retile_level(lv, tuple(int(v) for v in m.split(","))) for lv in H.levels], 2)
    dls = [capi.DevLevel(ctx, lv) for lv in T.levels]
    st, wk, ou, keep = [], [], [], []
    with torch.cuda.stream(stream):
        for li, (lv, dl) in enumerate(zip(T.levels, dls)):
            off, cs, tot = mf_layout(lv.boxes, ncomp, 2)
            t = torch.zeros(tot, dtype=torch.float64, device=dev)
            bench.fill_level_on_device(torch, lv, t, 1, 2, off, cs, dev, 177 + li)
            for b in range(lv.nboxes if ncomp > 1 else 0):
                nz, ny, nx = lv.box_shape(b, 2)
                n = nz * ny * nx
                for c in range(1, ncomp):
                    t[off[b] + c * cs[b]: off[b] + c * cs[b] + n] = (1.0 + 0.1 * c) * t[off[b]: off[b] + n] + 3.0 * c
            keep.append(t)
            st.append(capi.DevMF(ctx, dl, ncomp, 2, t.data_ptr())); wk.append(capi.DevMF(ctx, dl, 1, 2)); ou.append(capi.DevMF(ctx, dl, 8 * nbatch, 0))
    stream.synchronize()
    run = lambda: capi.gradcurv_run_comps2(ctx, st, 0, ncomp, bc, P, wk, ou, 0, nbatch)
    run(); ctx.sync()
    assert ctx.bc_errors() == 0
    ctx.profile_enable(True)
    for tag in range(1, 9):
        ctx.profile_read(tag, reset=True)
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        run()
    ctx.sync()
    ms = (time.perf_counter() - t0) / reps * 1e3
    fam = {name: round(ctx.profile_read(tag, reset=True)[1] / reps, 3) for tag, name in ((1, "sweep"), (2, "fixup"), (3, "fillb"), (4, "applybc"))}
    ctx.profile_enable(False)
    narrow = []
    for lv in T.levels:
        w = lv.boxes[:, 3] - lv.boxes[:, 0] + 1
        n = np.prod(lv.boxes[:, 3:].astype(np.int64) - lv.boxes[:, :3] + 1, axis=1)
        narrow.append(round(float(n[w <= 32].sum() / n.sum()), 3))
    print(f"{case} tiling {m}: boxes {[lv.nboxes for lv in T.levels]} share<=32wide {narrow} {ms:.3f} ms  frac {cells * 72 / (ms * 1e-3) / 8e12:.3f}  events {fam}  kernel {ctx.lib.pa_sweep_kernel_name(ctx.h).decode()}", flush=True)
    for x in st + wk + ou:
        x.close()
    for d in dls:
        d.close()
    del keep, st, wk, ou, dls
    torch.cuda.empty_cache()
