import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

ge.load_package()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure)."""
    from oracle import orc as _orc
    _orc.lib()
    return _orc


@pytest.fixture(scope="session")
def capi():
    from malio_amd import capi as _capi
    if not os.path.exists(_capi.LIB_PATH):
        ge.build()
    _capi.lib()
    return _capi


@pytest.fixture(scope="session")
def scenes():
    from malio_amd import scenes as _s
    return _s


def fused_from_rows(r):
    """Reference accumulation esekfom.hpp:621-635 from oracle rows: (HtRinvH, HtRinvh)."""
    Rc = np.where(r["R"] < 1e-4, 1e-3, r["R"])
    HT = r["h_x"].T / Rc
    return HT @ r["h_x"], HT @ r["h"]
