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


# The library has no tuning state; tests change kernels through a Python-side test Config (snipper_amd/_lib.py) that is
# dropped after EVERY test, so the order in which files are collected cannot decide which kernels a later test exercises.
@pytest.fixture(autouse=True)
def _library_defaults():
    yield
    try:
        from snipper_amd import _lib
        _lib.reset_config()
    except Exception:
        pass
