"""GPU parity: streamline tracer (pa_stream_trace, partStream.cpp / StreamPC.cpp) vs the oracle, bit for bit:
same FAB choice (lazy global re-assignment), same trilinear/RK4 operation order."""
import numpy as np
import pytest

from peleanalysis_amd import capi
from peleanalysis_amd.hierarchy import MultiFab, nested_hierarchy, fill_analytic, field_flame

pytestmark = pytest.mark.gpu


def _vfield_dev(ctx, H, vhost):
    dls = [capi.DevLevel(ctx, lv) for lv in H.levels]
    return dls, [capi.DevMF.from_host(ctx, dl, v) for dl, v in zip(dls, vhost)]


@pytest.mark.parametrize("ngrow,hrk", [(3, 0.4), (2, 0.1)])
def test_stream_matches_oracle(ctx, oracle, ngrow, hrk):
    H = nested_hierarchy(32, 3, 16, is_per=(0, 0, 0))
    fields = []
    for lv in H.levels:
        m = MultiFab(lv, 3, 0)
        # a swirling, non-linear field: rotation about the centre + the gradient direction of the flame kernel
        fill_analytic(m, 0, lambda x, y, z: -(y - 0.5) + 0.3 * np.sin(7 * z) + 0 * x)
        fill_analytic(m, 1, lambda x, y, z: (x - 0.5) + 0.2 * np.cos(5 * x) + 0 * y + 0 * z)
        fill_analytic(m, 2, lambda x, y, z: 0.25 * np.sin(6 * x) * np.cos(4 * y) + 0 * z)
        fields.append(m)
    v = oracle.stream_field(H.levels, fields, (0, 1, 2), MultiFab, ngrow=ngrow)
    rng = np.random.default_rng(9)
    seeds = 0.5 + 0.36 * (rng.random((200, 3)) - 0.5)
    nsteps, dt = 120, hrk / 128
    want, wred = oracle.stream_trace(H.levels, v, seeds, nsteps, dt)
    dls, dv = _vfield_dev(ctx, H, v)
    got, gred = capi.stream_trace(ctx, dv, 0, seeds, nsteps, dt)
    assert gred == wred and wred >= 1
    assert np.array_equal(got.view(np.int64), want.view(np.int64))


def test_stream_device_ghost_fill_and_errors(ctx, oracle):
    """the vector field prepared on the device like the tool does (fill_boundary + piecewise-constant FillPatch)
    equals the oracle's; too few ghost layers for the step -> the reference's 'bad RK' becomes an error code"""
    H = nested_hierarchy(16, 2, 8, is_per=(0, 0, 0))
    fields = []
    for lv in H.levels:
        m = MultiFab(lv, 3, 0)
        for c in range(3):
            fill_analytic(m, c, lambda x, y, z, c=c: field_flame(x, y, z, c) * 1e-3 - 1.0)
        fields.append(m)
    ng = 3
    want = oracle.stream_field(H.levels, fields, (0, 1, 2), MultiFab, ngrow=ng)
    dls = [capi.DevLevel(ctx, lv) for lv in H.levels]
    dv = []
    for l, (lv, dl) in enumerate(zip(H.levels, dls)):
        h = MultiFab(lv, 3, ng)
        for b in range(lv.nboxes):
            h.valid(b)[:] = fields[l].valid(b)
        d = capi.DevMF.from_host(ctx, dl, h)
        ctx.check(ctx.lib.pa_fill_boundary(ctx.h, d.h, 0, 3, ng))
        if l > 0:
            ctx.check(ctx.lib.pa_fillpatch_two_levels(ctx.h, d.h, dv[l - 1].h, 0, 3, ng, 2, 0))
        dv.append(d)
    ctx.sync()
    for l in range(H.nlev):
        assert np.array_equal(dv[l].download().data.view(np.int64), want[l].data.view(np.int64))
    seeds = np.array([[0.5, 0.5, 0.5], [0.4, 0.55, 0.6]])
    got, _ = capi.stream_trace(ctx, dv, 0, seeds, 30, 0.2 / 32)
    ref, _ = oracle.stream_trace(H.levels, want, seeds, 30, 0.2 / 32)
    assert np.array_equal(got.view(np.int64), ref.view(np.int64))
    with pytest.raises(capi.PaError, match="bad RK"):
        capi.stream_trace(ctx, dv, 0, seeds, 30, 2.5 / 16)  # a step of 2.5 coarse cells with nGrow = 3: leaves the FAB between checks
