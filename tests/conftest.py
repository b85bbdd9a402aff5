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


def _start_heartbeat():
    """On the GPU box a runner that sees no output for minutes takes the run for hung; a cold box can spend that long paging
    libraries in.  A line every 30 s in gpurun_out/ says the session is alive; a test that really blocks is ended by the
    watchdog above."""
    root = os.environ.get("GRAFT_REPO_ROOT")
    if not root:
        return
    import threading
    import time
    path = os.path.join(root, "gpurun_out", "pytest_heartbeat.txt")

    def beat():
        t0 = time.time()
        while True:
            try:
                os.makedirs(os.path.dirname(path), exist_ok=True)
                with open(path, "w") as f:
                    f.write(f"pytest pid {os.getpid()} alive, {time.time() - t0:.0f} s\n")
            except OSError:
                pass
            time.sleep(30)
    threading.Thread(target=beat, daemon=True).start()


_start_heartbeat()
