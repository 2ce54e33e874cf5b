import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu through gpurun)")
    # the page ring's service gives up when the host stops calling for this long: a test process that sits in an oracle
    # computation on a slow box is alive, not gone (the give-up test sets its own value)
    os.environ.setdefault("PBSGPU_RING_IDLE_TIMEOUT_S", "90")


@pytest.fixture(scope="session")
def O():
    """The CPU parity oracle (test infrastructure only)."""
    from oracle import oracle

    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def gpu_lib():
    from pbs_plus_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    L = _lib.lib()
    if L.pbsgpu_device_count() <= 0:
        pytest.fail("libpbsgpu sees no HIP device: the GPU tests have no CPU fallback to hide behind")
    return L
