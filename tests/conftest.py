import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU must fail loudly, not skip silently; plain runs skip gpu tests.
    if _has_gpu():
        return
    selected = config.getoption("-m") or ""
    if "gpu" in selected and "not gpu" not in selected:
        return
    skip = pytest.mark.skip(reason="no GPU here (hipcc cross-compiles only); run with -m gpu on MI355X")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
