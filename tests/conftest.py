import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

from __graft_entry__ import load_package  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """The tests that start REAL multi-rank RCCL groups (tests/test_gpu_dp_native.py, world > 1) have never met a box with
    more than one GPU: they go last, so that under `-x` a first failure there cannot hide the result of anything else."""
    def multi_rank(item):
        return "test_gpu_dp_native" in item.nodeid and not any(t in item.nodeid for t in ("[1-", "world1", "one_rank"))
    items[:] = [i for i in items if not multi_rank(i)] + [i for i in items if multi_rank(i)]


@pytest.fixture(scope="session")
def pkg():
    return load_package()


def _gpu_present():
    try:
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        n = ctypes.c_int(0)
        return hip.hipGetDeviceCount(ctypes.byref(n)) == 0 and n.value > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu(pkg):
    """GPU tests fail loudly (no fallback) if the HIP library or the device is missing."""
    pkg.capi.load()
    assert _gpu_present(), "no HIP device visible: -m gpu tests must run on the MI355X box"
    return True
