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
    from peleanalysis_amd import capi
    c = capi.Context(0)
    yield c
    c.close()


@pytest.fixture(params=["separable", "exact"])
def filter_mode(request):
    """The two forms of Filter::apply_filter (filterPlt.cpp:217): the separable default (<= 1e-12 * Linf of the oracle,
    SURVEY 8d metric) and PA_FILTER_EXACT=1 = the reference's tap order, bit for bit.  The library reads the switch per launch."""
    old = os.environ.get("PA_FILTER_EXACT")
    if request.param == "exact":
        os.environ["PA_FILTER_EXACT"] = "1"
    else:
        os.environ.pop("PA_FILTER_EXACT", None)
    yield request.param
    if old is None:
        os.environ.pop("PA_FILTER_EXACT", None)
    else:
        os.environ["PA_FILTER_EXACT"] = old
