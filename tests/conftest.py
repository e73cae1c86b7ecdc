import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu_ctx():
    # torch ships its own HIP runtime: let it find the device first, then libgarlic_hip joins
    # (the other order leaves torch without a device in this image)
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    torch.zeros(1, device="cuda")
    from garlic_amd import abi
    ctx = abi.Context(0)
    yield ctx
    ctx.close()
