import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip_lib_path():
    """Build (if stale) and return the path of the C-ABI library; hipcc cross-compiles without a GPU."""
    from cookietts_amd import build
    return build.build(verbose=False)


@pytest.fixture
def tuning(monkeypatch):
    """Flip a CTTS_* launch-shape knob for the rest of the test: the library reads its environment ONCE, so setting the
    variable alone does nothing - this sets it, makes the library re-read (ctts_tuning_reload) and checks that the knob
    is the one in effect (ctts_tuning_flags).  Everything is undone after the test."""
    from cookietts_amd import _lib

    class Knobs:
        def set(self, name, value="1"):
            monkeypatch.setenv(name, value)
            _lib.tuning_reload()
            assert _lib.tuning_active(name), name

        def clear(self, name):
            monkeypatch.delenv(name, raising=False)
            _lib.tuning_reload()
            assert not _lib.tuning_active(name), name
    yield Knobs()
    monkeypatch.undo()
    _lib.tuning_reload()


def rms_rel_err(a, b):
    import numpy as np
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.sqrt(np.mean((a - b) ** 2)) / np.sqrt(np.mean(b ** 2)))
