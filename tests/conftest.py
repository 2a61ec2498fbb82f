import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "earthkit-meteo_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ek():
    """The product package on a real GPU.  A machine without any AMD GPU device node skips the `-m gpu`
    tests; a machine WITH one (the GPU box) must load the HIP library and see the device -- anything else
    is a failure, never a silent skip or fallback."""
    if not os.path.exists("/dev/kfd"):
        pytest.skip("no AMD GPU on this machine (/dev/kfd missing)")
    import ekm_hip
    import ekm_hip.vertical  # noqa: F401

    assert ekm_hip.device_count() >= 1, "a GPU device node exists but HIP sees no device"
    return ekm_hip
