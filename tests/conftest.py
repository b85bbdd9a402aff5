import os
import sys

import pytest

TESTS = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(TESTS)
for p in (ROOT, TESTS):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    import golden_util as gu
    return gu.oracle()


@pytest.fixture(scope="session")
def lib():
    """The product library; built on demand (hipcc cross-compiles without a GPU)."""
    from cuclark_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib.load()


@pytest.fixture(scope="session")
def db_dir(tmp_path_factory):
    return str(tmp_path_factory.mktemp("golden_db"))


@pytest.fixture(autouse=True)
def _hang_watchdog():
    """A test that blocks for more than 6 minutes (a GPU call that never returns) ends the session with the Python stacks on stderr
    instead of sitting there until the runner's own limit kills it without a word."""
    import faulthandler
    faulthandler.dump_traceback_later(400, exit=True)
    yield
    faulthandler.cancel_dump_traceback_later()
