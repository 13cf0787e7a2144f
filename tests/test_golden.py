"""Committed golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py from the oracle
driven by the reference's own ikd-Tree). CPU: the oracle with its INDEPENDENT k-d tree must reproduce
them; GPU: the HIP path, through the C ABI, must reproduce them."""
import glob
import os

import numpy as np
import pytest

from conftest import assert_P_close

HERE = os.path.dirname(os.path.abspath(__file__))
# scene fixtures (predict_chain.npz is the propagation-step pin: tests/test_predict.py)
CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(HERE, "golden", "*.npz"))
               if os.path.basename(p) != "predict_chain.npz")


def load(name, orc):
    z = np.load(os.path.join(HERE, "golden", name + ".npz"))
    prm = {k: z["params"][i] for i, k in enumerate(orc.PARAM_ORDER) if i < len(z["params"])}  # older fixtures: no `limit`
    for k in ("lid_num", "max_iteration", "extrinsic_est_en"):
        prm[k] = int(prm[k])
    off = np.concatenate([[0], np.cumsum(z["table_len"])])
    tables = [z["tables"][off[i]:off[i + 1]] for i in range(len(z["table_len"]))]
    return z, prm, tables


def check_pass(z, pre, M, hx, h, R, rows_tol):
    assert M == int(z[pre + "_M"])
    assert np.allclose(hx, z[pre + "_hx"], rtol=0, atol=rows_tol * max(1.0, np.abs(z[pre + "_hx"]).max()))
    assert np.allclose(h, z[pre + "_h"], rtol=0, atol=rows_tol)
    assert np.allclose(R, z[pre + "_R"], rtol=1e-12, atol=1e-18)


def test_fixtures_present():
    assert len(CASES) >= 4


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_golden(orc, name):
    z, prm, tables = load(name, orc)
    o = orc.Oracle(prm, threads=2, use_ref=False)  # independent k-NN provider
    o.map_build(z["map"])
    o.scan_set(z["scan"], tables, z["temporal_comp"])
    r = o.h_share_model(z["state0"], True)
    g = o.scan_get()
    assert np.array_equal(g["selected"], z["p1_selected"])
    assert np.array_equal(g["world"], z["p1_world"])
    assert np.array_equal(g["normvec"], z["p1_normvec"])
    assert np.array_equal(g["normal_y"], z["p1_normal_y"])
    check_pass(z, "p1", r["M"], r["h_x"], r["h"], r["R"], 1e-13)
    assert r["weight"] == pytest.approx(float(z["p1_weight"]), rel=1e-12)
    r2 = o.h_share_model(z["state2"], False)
    assert np.array_equal(o.scan_get()["selected"], z["p2_selected"])
    check_pass(z, "p2", r2["M"], r2["h_x"], r2["h"], r2["R"], 1e-13)
    o.scan_set(z["scan"], tables, z["temporal_comp"])
    u = o.update_iterated(z["state0"], z["P0"])
    assert u["passes"] == int(z["upd_passes"]) and u["searches"] == int(z["upd_searches"])
    assert np.allclose(u["state"], z["upd_state"], rtol=0, atol=1e-12)
    assert np.allclose(u["P"], z["upd_P"], rtol=1e-9, atol=1e-15)
    _, d2, cnt = o.knn(z["knn_q"])
    assert np.array_equal(d2, z["knn_d2"]) and np.array_equal(cnt, z["knn_cnt"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_reproduces_golden(capi, orc, name):
    z, prm, tables = load(name, orc)
    eng = capi.Engine(prm, device=0)
    eng.map_build(z["map"])
    assert eng.map_size() == z["map"].shape[0]
    eng.scan_set(z["scan"], tables, z["temporal_comp"])
    r = eng.measure(z["state0"], True, want_rows=True)
    g = eng.scan_get()
    # discrete outcomes and float32 stages: bit-exact
    assert np.array_equal(g["selected"], z["p1_selected"])
    assert np.array_equal(g["world"], z["p1_world"])
    sel = z["p1_selected"].astype(bool)
    assert np.array_equal(g["normvec"][sel], z["p1_normvec"][sel])
    assert np.array_equal(g["nearest"][sel][:, :, :3], z["p1_nearest"][sel])
    # double stages: trace(Sigma_p) is evaluated as a folded quadratic form -> 1e-10 relative
    assert np.allclose(g["normal_y"], z["p1_normal_y"], rtol=1e-6, atol=0)
    check_pass(z, "p1", r["M"], r["h_x"], r["h"], r["R"], 1e-11)
    assert r["w_loc"] == pytest.approx(float(z["p1_weight"]), rel=1e-10)
    Rc = np.where(z["p1_R"] < 1e-4, 1e-3, z["p1_R"])
    HtH = (z["p1_hx"].T / Rc) @ z["p1_hx"]
    Hth = (z["p1_hx"].T / Rc) @ z["p1_h"]
    assert np.allclose(r["HtRinvH"], HtH, rtol=0, atol=1e-11 * np.abs(HtH).max())
    assert np.allclose(r["HtRinvh"], Hth, rtol=0, atol=1e-11 * np.abs(Hth).max())
    r2 = eng.measure(z["state2"], False, want_rows=True)
    assert np.array_equal(eng.scan_get()["selected"], z["p2_selected"])
    check_pass(z, "p2", r2["M"], r2["h_x"], r2["h"], r2["R"], 1e-11)
    eng.scan_set(z["scan"], tables, z["temporal_comp"])
    u = eng.update_iterated(z["state0"], z["P0"])
    assert u["passes"] == int(z["upd_passes"]) and u["searches"] == int(z["upd_searches"]) and u["M"] == int(z["upd_M"])
    assert np.allclose(u["state"], z["upd_state"], rtol=0, atol=1e-9)
    assert_P_close(u["P"], z["upd_P"])
    _, d2, cnt = eng.nearest_search(z["knn_q"])
    inside = z["knn_d2"] <= 5.0  # exact inside the sqrt(5) m acceptance radius (laserMapping.cpp:587)
    assert np.array_equal(d2[inside], z["knn_d2"][inside])
