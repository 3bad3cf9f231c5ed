import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    if os.environ.get("SADBM_ABRT_TRACE") == "1":              # debugging aid: native backtrace on SIGABRT (tools/dbg/abrt_trace.c)
        import ctypes
        so = os.path.join(ROOT, "tools", "dbg", "libabrt_trace.so")
        if os.path.exists(so):
            ctypes.CDLL(so).abrt_trace_install()


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(autouse=True)
def _library_knobs_follow_the_environment():
    """libsadbm reads its SADBM_* tuning knobs once; tests that flip one (monkeypatch.setenv + oa.reload_env()) get the
    library re-synchronised with the restored environment afterwards (this fixture is set up before monkeypatch, so it
    is torn down after it)."""
    yield
    import opencv_amd.stereo as st
    if st._lib is not None:
        st._lib.sadbm_reload_env()
