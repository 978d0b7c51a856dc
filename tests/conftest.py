import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# generated synthetic graphs are kept for the session (the MAG shape is 0.8 GB and several tests use it)
os.environ.setdefault("GRANDPLUS_SYNTH_CACHE", os.path.join("/dev/shm" if os.path.isdir("/dev/shm") else "/tmp", "gp_synth_tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Make sure every native piece exists (no-op when the .so files are fresh)."""
    import __graft_entry__ as ge
    ge.build()
