import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")
    config.addinivalue_line("markers", "slow: tens of seconds of CPU (still part of the default CPU suite)")


@pytest.fixture(scope="session", autouse=True)
def _built_libraries():
    """The oracle (test infrastructure) is compiled on demand; the product library must already be built in-tree
    (python -m blub_amd.build / __graft_entry__.build) -- build it here only if hipcc is available."""
    from oracle.oracle import build_oracle
    build_oracle()
    from blub_amd import build as b
    if b.needs_build() and os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")):
        b.build()
    yield


def has_gpu():
    try:
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        n = ctypes.c_int(0)
        return hip.hipGetDeviceCount(ctypes.byref(n)) == 0 and n.value > 0
    except OSError:
        return False
