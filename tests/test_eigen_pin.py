"""Pin of the CPU oracle against the reference's own Eigen-typed code (oracle/ref_eigen).

tests/golden/eigen_pin.npz holds outputs of esti_plane<float>, evalPointUncertainty, the compound functions,
esekf::update_iterated_dyn_share_modified (on replayed rows) and esekf::predict, compiled UNMODIFIED from
/root/reference against Eigen 3 (recipe: oracle/ref_eigen/Makefile, generator: tests/golden/make_eigen_pin.py).
Eigen and Boost are not installed in the images this repository has been built in so far, so the file does not exist
yet and the pin test SKIPS: until it runs, "bit-exact" in this repository means bit-exact to the restatement.
Tolerances once it runs: plane flag exact, plane coefficients 4 float ulp (Eigen's packetised reductions add in a
different order, oracle/orc_geom.cpp:14-16), double results 1e-11 relative to the largest entry."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import eigen_io  # noqa: E402
import make_eigen_pin as mk  # noqa: E402


def test_container_round_trip(tmp_path):
    a = dict(x=np.arange(6, dtype=np.float32).reshape(2, 3), y=np.array([1.5, -2.0]), n=np.array([3, 4, 5], np.int32))
    p = str(tmp_path / "c.bin")
    eigen_io.write(p, a)
    b = eigen_io.read(p)
    assert list(b) == list(a) and all(np.array_equal(a[k], b[k]) and a[k].dtype == b[k].dtype for k in a)


def test_pin_inputs_and_oracle_side(orc):
    """The generator's inputs are deterministic and the oracle evaluates every pinned quantity on them (replayed update:
    pass 2 invalid, pass 3 through the M < n branch) - the half of the pin that can run without Eigen."""
    inp, passes, sc = mk.build_inputs()
    inp2, _, _ = mk.build_inputs()
    assert all(np.array_equal(inp[k], inp2[k]) for k in inp)
    out = mk.oracle_outputs(inp, passes, sc)
    assert out["upd_passes"][0] == 5 or out["upd_passes"][0] == 4  # ends at `done` (t > 1) or at the last iteration
    assert np.isfinite(out["upd_P_out"]).all() and np.isfinite(out["pred_P_out"]).all()
    assert out["plane_ok"].sum() > 0.5 * (inp["plane_pts"].shape[0] - 40)
    assert np.allclose(out["unc_cov"], np.transpose(out["unc_cov"], (0, 2, 1)), rtol=1e-12, atol=1e-15)


def test_oracle_matches_reference_eigen_build(orc):
    if not os.path.exists(mk.GOLDEN):
        pytest.skip("tests/golden/eigen_pin.npz absent: Eigen 3 / Boost were not available to build oracle/ref_eigen - "
                    "oracle parity vs the reference's Eigen code is UNPINNED")
    g = np.load(mk.GOLDEN)
    inp, passes, sc = mk.build_inputs()
    for k in inp:  # the golden file was made from the same inputs
        assert np.array_equal(inp[k], g["in_" + k]), k
    out = mk.oracle_outputs(inp, passes, sc)
    assert np.array_equal(out["plane_ok"], g["ref_plane_ok"])
    ok = out["plane_ok"].astype(bool)
    ulp = np.spacing(np.abs(g["ref_plane_pabcd"][ok]).astype(np.float32))
    assert (np.abs(out["plane_pabcd"][ok] - g["ref_plane_pabcd"][ok]) <= 4 * ulp).all()
    for k in ("plane_cov", "unc_cov", "comp_out", "comp_inv_out", "upd_state_out", "pred_state_out"):
        ref = g["ref_" + k]
        assert np.abs(out[k] - ref).max() <= 1e-11 * max(1.0, np.abs(ref).max()), k
    assert out["upd_passes"][0] == g["ref_upd_passes"][0]
    from conftest import assert_P_close
    assert_P_close(out["upd_P_out"], g["ref_upd_P_out"], rel=1e-6)
    for a, b in zip(out["pred_P_out"], g["ref_pred_P_out"]):
        assert_P_close(a, b, rel=1e-9)
