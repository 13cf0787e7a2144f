"""Resident front end (rows a13 -> f-2 -> scan upload without host round trips): malio_undistort_resident +
malio_scan_set_resident must produce exactly the scan, and therefore exactly the measurement update, that the
host-buffer calls produce when chained the way the reference chains them (IMU_Processing.hpp:475-507,
laserMapping.cpp:966-983)."""
import numpy as np
import pytest


def _traj(scenes, t0, dur=0.32, rate=200.0):
    ts = t0 + np.arange(0, dur, 1.0 / rate)
    w, v = np.array([0.02, -0.01, 0.03]), np.array([0.15, 0.02, -0.01])      # slow: the scan stays on the map
    return np.array([[t, *(v * (t - t0)), *scenes.q_from_rotvec(w * (t - t0))] for t in ts])


@pytest.mark.gpu
@pytest.mark.parametrize("L", [1, 3])
def test_resident_front_end_equals_host_chain(capi, scenes, L):
    sc = scenes.make_scene(seed=31 + L, N=12000, Nmap=80000, L=L)
    rng = np.random.default_rng(5)
    t0 = 1671631987.6
    traj = _traj(scenes, t0)
    kt, kT = capi.spline_feed(traj)
    beg, end = t0 + 0.05, t0 + 0.15
    _, q_end, p_end = capi.spline_get_pose(kt, kT, end)
    imu_t = traj[::2, 0].copy()
    cp = int(np.searchsorted(imu_t, end, side="right"))
    st = scenes.unpack_state(sc["state_gt"], L)
    leaf = 0.4
    # raw clouds per LiDAR: the scene's scan points of that LiDAR, each repeated with jitter (so the voxel filter has
    # something to merge), with time offsets in `curvature`, sorted by time as the reference requires (:229-233)
    raws = []
    for l in range(L):
        base = sc["scan"][sc["scan"][:, 8] == l]
        raw = np.repeat(base, 3, axis=0).copy()
        raw[:, :3] += rng.normal(0, 0.05, size=(raw.shape[0], 3)).astype(np.float32)
        raw[:, 9] = np.sort(rng.uniform(0, (end - beg) * 1000.0, raw.shape[0])).astype(np.float32)
        raw[:, 5] = rng.uniform(0, 0.01, raw.shape[0]).astype(np.float32)
        raw[:, 8] = rng.uniform(0, 200, raw.shape[0]).astype(np.float32)    # reflectivity before undistortion
        raws.append(raw)

    host = capi.Engine(sc["params"])
    host.map_build(sc["map"])
    downs, ents_h = [], []
    for l in range(L):
        und, ent = host.undistort(raws[l], beg, kt, kT, st["offR"][l], st["offT"][l], q_end, p_end, imu_t, cp)
        ents_h.append(ent)
        d = host.voxel_downsample(und, leaf)
        d[:, 4] = d[:, 8]          # normal_x <- intensity   (laserMapping.cpp:974)
        d[:, 8] = l                # intensity <- num        (:975)
        downs.append(d)
    body_h = np.concatenate(downs, 0)  # *feats_down_body += *feats_down_vec[num]  (:982)
    host.scan_set(body_h, sc["tables"], sc["temporal_comp"])
    r_h = host.measure(sc["state0"], True, want_rows=True)
    s_h = host.scan_get()

    res = capi.Engine(sc["params"])
    res.map_build(sc["map"])
    res.scan_order(1)          # sort like the host-buffer chain does, so that sums can be compared bit for bit
    for l in range(L):
        ent, epts = res.undistort_resident(l, raws[l], beg, kt, kT, st["offR"][l], st["offT"][l], q_end, p_end, imu_t, cp)
        np.testing.assert_array_equal(ent, ents_h[l])
        und, _ = host.undistort(raws[l], beg, kt, kT, st["offR"][l], st["offT"][l], q_end, p_end, imu_t, cp)
        np.testing.assert_array_equal(epts, und[ent])
    body_r = res.scan_set_resident(leaf, sc["tables"], sc["temporal_comp"])
    np.testing.assert_array_equal(body_r, body_h)
    r_r = res.measure(sc["state0"], True, want_rows=True)
    s_r = res.scan_get()
    assert r_r["M"] == r_h["M"] and r_h["M"] > 0.3 * body_h.shape[0]
    for k in ("HtRinvH", "HtRinvh", "h_x", "h", "R"):
        np.testing.assert_array_equal(r_r[k], r_h[k])
    for k in ("selected", "res_last", "normal_y", "world", "nearest_cnt"):
        np.testing.assert_array_equal(s_r[k], s_h[k])
    # and the loop goes on: update + map_incremental on the resident scan
    u_r, u_h = res.update_iterated(sc["state0"], sc["P0"]), host.update_iterated(sc["state0"], sc["P0"])
    np.testing.assert_array_equal(u_r["state"], u_h["state"])
    wny = np.zeros(body_h.shape[0], np.float32)
    assert res.map_incremental(u_r["state"], True, wny) == host.map_incremental(u_h["state"], True, wny)
    # the resident clouds were consumed
    with pytest.raises(RuntimeError):
        res.scan_set_resident(leaf, sc["tables"], sc["temporal_comp"])
    # default order of a resident scan: as the voxel filter left it (no spatial sort). Same accepted points, same
    # per-point values; sums (and what is scaled by the weight derived from them) to rounding: another summation order.
    res.scan_order(0)
    for l in range(L):
        res.undistort_resident(l, raws[l], beg, kt, kT, st["offR"][l], st["offT"][l], q_end, p_end, imu_t, cp)
    body_k = res.scan_set_resident(leaf, sc["tables"], sc["temporal_comp"])
    np.testing.assert_array_equal(body_k, body_h)
    host.scan_set(body_h, sc["tables"], sc["temporal_comp"])
    r_s = host.measure(sc["state0"], True, want_rows=True)
    r_k = res.measure(sc["state0"], True, want_rows=True)
    assert r_k["M"] == r_s["M"]
    np.testing.assert_array_equal(r_k["R"], r_s["R"])        # per-point values do not depend on the order ...
    for k in ("h_x", "h"):                                    # ... the rows carry the localization weight, which does
        assert np.allclose(r_k[k], r_s[k], rtol=1e-13, atol=0)
    assert np.abs(r_k["HtRinvH"] - r_s["HtRinvH"]).max() <= 1e-12 * np.abs(r_s["HtRinvH"]).max()
    assert np.abs(r_k["HtRinvh"] - r_s["HtRinvh"]).max() <= 1e-12 * np.abs(r_s["HtRinvh"]).max()
    np.testing.assert_array_equal(res.scan_get()["selected"], host.scan_get()["selected"])
    with pytest.raises(RuntimeError):
        res.scan_order(7)
    res.scan_order(1)
    # same chain with the caller's clouds in page-locked memory (malio_host_alloc): same bytes, fewer stalls
    pins = [capi.PinnedArray(r.shape, np.float32) for r in raws]
    pout = capi.PinnedArray((sum(r.shape[0] for r in raws), 12), np.float32)
    for l in range(L):
        pins[l].array[:] = raws[l]
        ent, _ = res.undistort_resident(l, pins[l].array, beg, kt, kT, st["offR"][l], st["offT"][l], q_end, p_end, imu_t, cp)
        np.testing.assert_array_equal(ent, ents_h[l])
    body_p = res.scan_set_resident(leaf, sc["tables"], sc["temporal_comp"], out=pout.array)
    np.testing.assert_array_equal(body_p, body_h)


@pytest.mark.gpu
def test_host_alloc_errors(capi):
    import ctypes as C
    lib = capi.lib()
    p = C.c_void_p()
    assert lib.malio_host_alloc(C.c_size_t(0), C.byref(p)) == capi.ERR_BAD_ARG
    assert lib.malio_host_alloc(C.c_size_t(16), None) == capi.ERR_BAD_ARG
    assert lib.malio_host_free(None) == capi.OK
    a = capi.PinnedArray((1000, 12), np.float32)
    a.array[:] = 1.5
    assert float(a.array.sum()) == 1.5 * 12000
