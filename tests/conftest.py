import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True)
def _library_options_restored():
    """the library's options travel in ONE environment variable (FH_DEBUG, include/finch_hip.h); a test that sets some
    (finch_rs_amd.debug_set) leaves the variable as it found it"""
    old = os.environ.get("FH_DEBUG")
    yield
    if old is None:
        os.environ.pop("FH_DEBUG", None)
    else:
        os.environ["FH_DEBUG"] = old


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """which BASELINE configurations ran at their full size (tests/test_gpu_full_size.py): a driver's tail shows it without -rs"""
    mod = sys.modules.get("test_gpu_full_size")
    if mod is None or "not gpu" in (config.getoption("-m") or ""):
        return
    ran = getattr(mod, "FULL_RAN", [])
    terminalreporter.write_line("BASELINE configurations sketched at their FULL size and held against the oracle: %s" %
                                (", ".join(ran) if ran else "none (the _scaled twins ran; FH_REQUIRE_FULL=1 makes that a failure)"))
