import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: exhaustive sweep, minutes on CPU")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O

    O.build()
    return O


@pytest.fixture(scope="session")
def ce():
    """The product binding; building is part of the session so CPU-only runs check that it compiles."""
    b = importlib.import_module("codec-eval_amd.build")
    b.build()
    import codec_eval_amd

    return codec_eval_amd


@pytest.fixture(scope="session")
def workloads():
    return importlib.import_module("codec-eval_amd.workloads")


@pytest.fixture(scope="session")
def gpu_ctx(ce):
    """One context for the whole GPU session.  Fails loudly (no skip) if the HIP path is unusable."""
    n = ce.device_count()
    assert n > 0, "no HIP device visible: the -m gpu tests must run on the GPU box"
    ctx = ce.Context(0)
    yield ctx
    ctx.close()
