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


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure; built on demand with gcc)."""
    from oracle import oracle as O

    O.build()
    return O


@pytest.fixture
def rows_form(monkeypatch):
    """The frame-ordered form of the window's row kernel (SAF_WIN_FORM=rows): bit-identical to fusing frame after frame.  The
    default is the order-free form (sums), whose feature values agree within fp32 rounding (tests/test_sums_form.py)."""
    monkeypatch.setenv("SAF_WIN_FORM", "rows")
