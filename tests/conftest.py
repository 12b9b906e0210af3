import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def engine():
    """The HIP engine behind the C ABI.  GPU tests fail loudly (no fallback) if the library is missing."""
    import torch  # noqa: F401  (loads the ROCm runtime first so torch and libpinot_gpu.so share one libamdhip64)
    from pinot_amd.engine import Engine
    return Engine(device_id=0, time_kernels=True)
