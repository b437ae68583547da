import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Both shared objects exist (building is the driver's build(); here we only make sure)."""
    import __graft_entry__ as g

    g.build()
    return True


@pytest.fixture(scope="session")
def cornell():
    from stratum_amd import scenes

    return scenes.cornell_box()
