import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture(scope="session", autouse=True)
def _heartbeat():
    """A few GPU tests spend minutes inside ONE float64 CPU oracle call (the 160^3 gradient parity test: ~5 min) while pytest -q
    prints nothing; a GPU box's watchdog takes 7 minutes of silence for a hang.  A daemon thread touches gpurun_out/.heartbeat every
    30 s for the length of the session (the directory the watchdog also looks at); nothing else depends on it."""
    import threading
    import time
    stop = threading.Event()
    path = os.path.join(os.environ.get("GRAFT_REPO_ROOT", ROOT), "gpurun_out")

    def beat():
        try:
            os.makedirs(path, exist_ok=True)
        except OSError:
            return
        t0 = time.time()
        while not stop.wait(30.0):
            try:
                with open(os.path.join(path, ".heartbeat"), "a") as f:
                    f.write(f"pytest session alive, {time.time() - t0:.0f} s\n")
            except OSError:
                return
    th = threading.Thread(target=beat, daemon=True)
    th.start()
    yield
    stop.set()
