import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # every query the GPU tests run is first put to pg_query_check: the two must agree on what is declined (pinot_amd/engine.py)
    os.environ.setdefault("PINOT_GPU_ASSERT_QUERY_CHECK", "1")


def pytest_sessionstart(session):
    """The native libraries are build artefacts (git-ignored): a fresh checkout that runs the tests before
    `__graft_entry__.build()` gets them built here (hipcc cross-compiles gfx950 without a GPU; make is a no-op when they are current).
    On the GPU box the prebuilt files travel with the snapshot and there may be no compiler: nothing is built when all of them exist."""
    libs = [os.path.join(ROOT, "pinot_amd", "csrc", "libpinot_gpu.so"), os.path.join(ROOT, "pinot_amd", "csrc", "libpinot_host.so"),
            os.path.join(ROOT, "oracle", "_build", "libpinot_oracle.so"), os.path.join(ROOT, "jni", "libpinot_gpu_marshal.so"),
            os.path.join(ROOT, "jni", "libpinot_gpu_jni_fake.so")]
    if not all(os.path.exists(p) for p in libs):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def engine():
    """The HIP engine behind the C ABI.  GPU tests fail loudly (no fallback) if the library is missing."""
    import torch  # noqa: F401  (loads the ROCm runtime first so torch and libpinot_gpu.so share one libamdhip64)
    from pinot_amd.engine import Engine
    return Engine(device_id=0, time_kernels=True)
