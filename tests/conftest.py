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
    O.lib()
    return O


@pytest.fixture(scope="session")
def gpu_ctx():
    # torch ships its own HIP runtime: when a test also needs torch device tensors,
    # torch must initialise the GPU BEFORE libbuildingsegment_hip.so (linked against
    # /opt/rocm) is loaded -- the same order bench.py uses
    try:
        import torch
        if torch.cuda.is_available():
            torch.zeros(1, device="cuda")
    except Exception:
        pass
    from buildingsegment_amd import api
    ctx = api.Context(0)
    yield ctx
    ctx.close()
