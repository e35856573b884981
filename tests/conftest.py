import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "pg-alignments-gan_amd")
for p in (ROOT, PKG, os.path.join(ROOT, "oracle"), os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _poisoned_allocator(request):
    """Every GPU test starts with NaN in the memory its torch.empty tensors will get (see helpers.poison_gpu_cache)."""
    if "gpu" in request.keywords:
        import torch
        if torch.cuda.is_available():
            from helpers import poison_gpu_cache
            poison_gpu_cache()
    yield
