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


def _gpu_count():
    """malio_device_count: the library's own HIP runtime is asked (a second copy of libamdhip64 loaded through ctypes
    would initialise a second runtime in the process, and the one that comes second finds no device)."""
    try:
        from malio_amd import capi as _c
        return int(_c.lib().malio_device_count())
    except Exception:
        return 0


def pytest_collection_modifyitems(config, items):
    """A plain `pytest tests` on a machine without a GPU: the `gpu` tests are skipped (with the reason) instead of failing
    one by one inside malio_create. On the GPU box nothing is skipped; `-m gpu` (the driver's GPU tier) never skips -
    without a device those tests fail, so a box that lost its GPU cannot pass as green - and MALIO_REQUIRE_GPU=1 turns a
    missing device into a usage error for any selection."""
    if not any("gpu" in it.keywords for it in items):
        return
    if _gpu_count() > 0:
        return
    if os.environ.get("MALIO_REQUIRE_GPU") == "1":
        raise pytest.UsageError("MALIO_REQUIRE_GPU=1 but hipGetDeviceCount reports no device")
    if "gpu" in (config.getoption("markexpr") or "") and "not gpu" not in config.getoption("markexpr"):
        return  # `-m gpu` asks for the GPU tests by name: without a device they FAIL (loudly), they are not skipped
    skip = pytest.mark.skip(reason="no gfx950 device (hipGetDeviceCount == 0): the HIP path has no CPU fallback")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


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


def assert_P_close(P, Q, rel=2e-3, HtH=None, P0=None):
    """Posterior covariances, compared element-wise against the LARGER of two scale-aware bounds:
      correlation scale      |dP_ij| <= rel * sqrt(Q_ii Q_jj)
      rounding of the reference's own formula (needs HtH, P0): the reference forms K_x = P_inv[:, :C] * HtH and
      P = L - K_x P (esekfom.hpp:637,714); with |HtH| ~ 1e11 (lever arm^2 x 1e5 points / R) against |P_inv| ~ 1e-7
      the products cancel over ~11 digits, so any implementation of that formula - the reference's Eigen build
      included - carries an absolute error ~ 64 ulp * (|P_inv| |HtH| |P0|)_ij in the pose/rotation cross block."""
    dg = np.sqrt(np.abs(np.diag(Q)))
    bound = rel * np.outer(dg, dg) + 1e-18
    if HtH is not None and P0 is not None:
        C = HtH.shape[0]
        bound = np.maximum(bound, 64 * 2.3e-16 * (np.abs(Q[:, :C]) @ np.abs(HtH) @ np.abs(P0[:C, :])))
    bad = np.abs(P - Q) > bound
    assert not bad.any(), (int(bad.sum()), float((np.abs(P - Q) / bound).max()))


def exact_ties(gs, os_, limit=8, at_search_state=True):
    """Queries whose Nearest_Points differ between two searches ONLY by an exact tie of float distances. ikdtree.Nearest_Search
    (ikd_Tree.cpp:426-461,1073-1255) orders two map points at the same float d2 by whatever its traversal and its heap leave
    (tools/tie_census.py: in 1.2 M queries of BASELINE config 5 five such pairs, four in ascending map index, one descending); the
    engine orders them by (d2, slot in its map array), and the slots are kept in cell order since round 6. Such a query - about
    four in a million - has the same five DISTANCES either way (or the same four and an equally distant fifth) and is compared on
    those; everything derived from the order of its five points (the plane's last bits, hence its gates) is left out of the
    per-point comparisons. Returns the boolean mask; fails when two searches differ in any other way, or more often than `limit`.
    at_search_state = False: gs / os_ were read after a REUSE pass (the neighbours are the last search pass', the world points
    are not): the distances cannot be formed, only the count is checked (the search pass before it was compared in full)."""
    differ = (gs["nearest"][:, :, :3] != os_["nearest"][:, :, :3]).any(axis=(1, 2))
    w = gs["world"][differ][:, None, :].astype(np.float32)

    def d2(n):
        d = w - n[:, :, :3]
        return (d[:, :, 0] * d[:, :, 0] + d[:, :, 1] * d[:, :, 1]) + d[:, :, 2] * d[:, :, 2]  # calc_dist, ikd_Tree.cpp:1697
    if not differ.any():
        return differ
    assert np.array_equal(gs["world"], os_["world"]) and np.array_equal(gs["nearest_cnt"], os_["nearest_cnt"])
    if at_search_state:
        dg, do = d2(gs["nearest"][differ]), d2(os_["nearest"][differ])
        assert np.array_equal(dg, do), "neighbours differ by more than an exact tie of distances"
    assert differ.sum() <= limit, int(differ.sum())
    return differ
