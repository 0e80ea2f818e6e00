import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip_lib():
    """The HIP library, built if needed. GPU tests fail loudly (no skip, no CPU fallback) when no device is visible."""
    from open_ludwig_amd import _lib, build
    build.build_library()
    lib = _lib.load()
    return lib


@pytest.fixture(scope="session")
def gpu():
    # torch first, the library second - the order bench.py and the workers use. torch bundles its own HIP / HSA runtime (no SONAME, so the
    # loader does not share it with /opt/rocm's, which libludwig_hip.so links): brought up AFTER the library has been stepping for minutes
    # in the same process, torch found "No HIP GPUs" (round 3, tests/test_rccl_loopback.py in-process); brought up first, both work.
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass
    from open_ludwig_amd import _lib, build
    build.build_library()
    _lib.load()
    n = _lib.device_count()
    assert n >= 1, "GPU test selected but libludwig_hip.so sees no HIP device"
    return 0
