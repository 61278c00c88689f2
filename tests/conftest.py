import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure only)."""
    from oracle import pyoracle
    pyoracle.build()
    return pyoracle


@pytest.fixture(scope="session")
def capi():
    from monica_amd import _capi
    _capi.lib()
    return _capi


@pytest.fixture(autouse=True)
def _resource_trace(request):
    """MNC_TEST_TRACE=<file>: open descriptors, threads and resident memory of the test process after every test
    (a leak across the suite shows as a slope)."""
    yield
    path = os.environ.get("MNC_TEST_TRACE")
    if path:
        import threading
        with open("/proc/self/statm") as f:
            rss = int(f.read().split()[1]) * os.sysconf("SC_PAGE_SIZE") >> 20
        with open(path, "a") as f:
            f.write(f"{len(os.listdir('/proc/self/fd')):5d} fds {threading.active_count():4d} threads {rss:7d} MB  {request.node.nodeid}\n")
