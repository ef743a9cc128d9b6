import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# Order of the run (the driver uses -x): the static name check first, then every test that compares against an output of the
# reference itself (tests/golden/*.npz), cheap ones before the full-length ones, then oracle / property tests, and the heavy
# full-size property runs last - a slip late in the run must cost property checks, never the parity evidence.
_REFERENCE_PINNED = ("reference_golden", "match_golden", "match_reference", "matches_long_goldens",
                     "where_the_reference_does", "options_match_reference", "goldens_configs")
_HEAVY = ("full_size", "full_length", "soak", "two_streams_share_the_chip")


def _run_tier(item):
    name = item.nodeid
    if "test_static_names" in name:
        return 0
    heavy = any(k in name for k in _HEAVY)
    if any(k in name for k in _REFERENCE_PINNED):
        return 2 if heavy else 1
    return 4 if heavy else 3


def pytest_collection_modifyitems(config, items):
    items.sort(key=_run_tier)                                                  # stable: file order kept inside a tier


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
