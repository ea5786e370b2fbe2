import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_terminal_summary(terminalreporter):
    """The strict per-output relative errors behind every helpers.rel_err check (floor 1e-3 of the reference tensor's RMS
    instead of the RMS itself): the five largest, for the record."""
    try:
        import helpers
    except Exception:
        return
    log = sorted(helpers.STRICT_LOG, reverse=True)
    if not log:
        return
    terminalreporter.write_line(f"strict per-output relative error over {len(log)} tensor comparisons (floor = 1e-3 x RMS): "
                                f"median {log[len(log) // 2][0]:.2e}, worst five: " +
                                ", ".join(f"{v:.2e} at {w}" for v, w in log[:5]))
