import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def ctx():
    """HIP context on cuda:0 -- the product path; raises (no CPU fallback) without a GPU."""
    try:  # torch ships its own HIP runtime: it must be the first one a process loads, or torch later finds "No HIP GPUs" (a test file
        import torch  # noqa: F401  -- that uses torch after one that only used the library: test_gpu_leaks.py after test_gpu_sdf.py)
    except ImportError:
        pass
    from peleanalysis_amd import capi
    c = capi.Context(0)
    yield c
    c.close()


@pytest.fixture(params=["separable", "exact"])
def filter_mode(request):
    """The two forms of Filter::apply_filter (filterPlt.cpp:217): the separable default (<= 1e-12 * Linf of the oracle,
    SURVEY 8d metric) and PA_FILTER_EXACT=1 = the reference's tap order, bit for bit.  The library reads its switches once:
    pa_options_reload after every change."""
    from peleanalysis_amd import capi
    old = os.environ.get("PA_FILTER_EXACT")
    if request.param == "exact":
        os.environ["PA_FILTER_EXACT"] = "1"
    else:
        os.environ.pop("PA_FILTER_EXACT", None)
    capi.reload_options()
    yield request.param
    if old is None:
        os.environ.pop("PA_FILTER_EXACT", None)
    else:
        os.environ["PA_FILTER_EXACT"] = old
    capi.reload_options()


@pytest.fixture
def options(monkeypatch):
    """set PA_* switches for one test -- options(PA_NCG=0), options(PA_FUSED2=None) -- and restore them afterwards: the library
    reads its switches once (peleanalysis_amd/csrc/pa_internal.h: pa_options), so every change is followed by pa_options_reload"""
    from peleanalysis_amd import capi

    def setter(**kw):
        for k, v in kw.items():
            if v is None:
                monkeypatch.delenv(k, raising=False)
            else:
                monkeypatch.setenv(k, str(v))
        capi.reload_options()
    yield setter
    monkeypatch.undo()
    capi.reload_options()
