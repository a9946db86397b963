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


# Library knobs (csrc/msda_capi.hip) are process-wide: whatever a test changes is put back after EVERY test, so the order
# in which files are collected cannot decide which kernels a later test exercises.
_KNOB_DEFAULTS = {"near_radius": 6.0, "owner_tile_edge_big": 16, "owner_tile_edge_mid": 8, "owner_tile_edge_small": 4,
                  "owner_debug": 0, "owner_chunk": 64, "ln_bwd_blocks": 1536, "wgrad_wgs": 512, "owner_enable": 1}


@pytest.fixture(autouse=True)
def _library_defaults():
    yield
    try:
        from snipper_amd import _lib
        if _lib._lib is None:          # never loaded by this test: nothing to restore
            return
        _lib.set_policy(0)
        for k, v in _KNOB_DEFAULTS.items():
            _lib.set_param(k, v)
    except Exception:
        pass
