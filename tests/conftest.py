import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun / round-end driver)")


@pytest.fixture(scope="session")
def gsfm_ctx():
    """One libgsfm context per test session; fails loudly when the HIP library / GPU is missing."""
    from glomap_amd import _lib

    ctx = _lib.Context(-1)
    yield ctx
    ctx.close()


# ---- oracle/_ref: make its absence LOUD (Weak 9) -----------------------------------------------------------
# A third of the parity evidence compares against the reference's own sources compiled into oracle/_ref/*.so — git-ignored
# libraries built from /root/reference by `make -C oracle ref ref_solve ref_dropin ref_mapper`.  Where they are missing those tests skip;
# a green bar must not hide that: the session ends with a line that counts them, and where the reference tree EXISTS (the build
# container) a missing library is a build failure of the shims, reported as an error instead of a skip.
_REF_SKIPS = []


def pytest_runtest_logreport(report):
    if report.skipped and "oracle/_ref" in str(report.longrepr):
        _REF_SKIPS.append(report.nodeid)


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    if _REF_SKIPS:
        terminalreporter.write_line(
            f"[oracle/_ref] {len(_REF_SKIPS)} reference-code tests SKIPPED: libraries under oracle/_ref/ absent (built from /root/reference "
            f"by `make -C oracle ref ref_solve ref_dropin ref_mapper`; they travel with the gpurun snapshot)", red=True, bold=True)


def pytest_sessionfinish(session, exitstatus):
    if _REF_SKIPS and os.path.exists("/root/reference/glomap/scene/view_graph.cc") and exitstatus == 0:
        # the reference tree is here, so the libraries should have been built: their absence is a shim compile error
        session.exitstatus = 1
