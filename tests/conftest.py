import ctypes
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _gpu_count() -> int:
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        n = ctypes.c_int(0)
        if hip.hipGetDeviceCount(ctypes.byref(n)) != 0:
            return 0
        return n.value
    except OSError:
        return 0


HAVE_GPU = _gpu_count() > 0


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if HAVE_GPU:
        return
    skip = pytest.mark.skip(reason="no HIP device in this container (GPU tests run through gpurun)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture
def lab(monkeypatch):
    """Engines created inside the test bind librvb_test.so, the build that reads the tuning switches (RVD_CONV_STREAM,
    RVD_LINKAGE_MB, ...) from the environment; the product library ignores them (csrc/common.h lab_env)."""
    monkeypatch.setenv("RVB_LAB", "1")


@pytest.fixture(scope="session")
def lib():
    from reverb_amd import _lib
    return _lib.load_test()        # librvb_test.so: the product's objects + the rvb_test_* hooks (csrc/test_api.h)
