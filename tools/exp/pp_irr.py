import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from peleanalysis_amd import capi
from peleanalysis_amd.hierarchy import tagged_hierarchy, field_flame, mf_layout
import bench
dev = torch.device("cuda:0")
stream = torch.cuda.Stream(device=dev)
ctx = capi.Context(0, stream.cuda_stream) if os.environ.get('PP_TORCH_STREAM') else capi.Context(0)
H = tagged_hierarchy(512, 3, lambda x, y, z: field_flame(x, y, z, 0), bf=16, max_box=128, base_box=128, frac=(0.08, 0.16), is_per=(1, 1, 0))
dls = [capi.DevLevel(ctx, lv) for lv in H.levels]
st, wk, ou, keep = [], [], [], []
for li, (lv, dl) in enumerate(zip(H.levels, dls)):
    off, cs, tot = mf_layout(lv.boxes, 1, 2)
    t = torch.zeros(tot, dtype=torch.float64, device=dev)
    bench.fill_level_on_device(torch, lv, t, 1, 2, off, cs, dev, 177 + li)
    keep.append(t)
    st.append(capi.DevMF(ctx, dl, 1, 2, t.data_ptr())); wk.append(capi.DevMF(ctx, dl, 1, 2)); ou.append(capi.DevMF(ctx, dl, 8, 0))
torch.cuda.synchronize()
bc = capi.bc_from_flags((1, 1, 0))
P = capi.curv_params(prog_min=300.0, prog_max=2003.0, threshold=None, fused=False)
if os.environ.get('PP_CLONE'):
    keep_f = [torch.empty(int(lv.ncells) * 8 + 4096 * lv.nboxes, dtype=torch.float64, device=dev) for lv in H.levels]
    torch.cuda.synchronize()
for r in range(5):
    t0 = time.perf_counter()
    capi.gradcurv_run(ctx, st, 0, bc, P, wk, ou, 0)
    ctx.sync()
    print("pass-by-pass call %d: %.2f ms" % (r, (time.perf_counter() - t0) * 1e3), flush=True)
