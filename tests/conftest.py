import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference (build container only)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    import refload
    skip_ref = pytest.mark.skip(reason="/root/reference not present (GPU box)")
    skip_gpu = pytest.mark.skip(reason="no GPU visible")
    gpu = has_gpu()
    for item in items:
        if "reference" in item.keywords and not refload.available():
            item.add_marker(skip_ref)
        if "gpu" in item.keywords and not gpu:
            item.add_marker(skip_gpu)
