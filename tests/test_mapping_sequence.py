"""The whole per-scan loop over a SEQUENCE of scans (laserMapping.cpp:985-1060): measurement update at the predicted
state, map_incremental at the posterior, box deletion behind the vehicle - with the map evolving under the search
structure from scan to scan. GPU engine vs oracle (reference ikd-Tree + restatements), step by step.

Both sides are fed the ORACLE's posterior as the basis of the next prediction, so every step compares like with like
(no drift between the two runs), while the GPU map is only ever changed by the GPU's own map maintenance."""
import numpy as np
import pytest


def _world(scenes, rng, half_w=60.0):
    p = scenes._surface_voxels("city", half_w, rng, scenes.SURFACE_SHIFT)
    pts = np.zeros((p.shape[0], 12), np.float32)
    pts[:, :3] = p.astype(np.float32)
    pts[:, 3] = 1.0
    pts[:, 5] = 0.001
    return pts


def _scan_from(scenes, rng, world, pos, rot, ext_q, ext_t, tc, N, L, det_range, n_table):
    """N points of the world surfaces around `pos`, re-drawn inside their voxels, expressed in the LiDAR frames."""
    origin = scenes.SURFACE_SHIFT
    rel = world[:, :3].astype(np.float64) - pos[None, :]
    near = np.nonzero(np.einsum("ij,ij->i", rel, rel) < det_range ** 2)[0]
    sel = rng.permutation(near)[:N]
    base = world[sel, :3].astype(np.float64)
    v = 0.5
    pw = (np.floor((base - origin) / v) * v + origin) + rng.uniform(0, v, (sel.size, 3))
    axis = np.argmin(scenes._offplane_hint("city", base - origin), axis=1)
    idx = np.arange(sel.size)
    pw[idx, axis] = base[idx, axis] + rng.normal(0, 0.02, sel.size)
    R = scenes.q_to_R(rot)
    b = (pw - pos[None, :]) @ R
    lid = rng.choice(L, size=sel.size, p=[0.65, 0.35])
    pb = np.zeros_like(b)
    for l in range(L):
        m = lid == l
        Rl = scenes.q_to_R(ext_q[l])
        if l == 0:
            pb[m] = (b[m] - ext_t[0][None, :]) @ Rl
        else:
            Rtc, ttc = scenes.q_to_R(tc[l - 1, 0:4]), tc[l - 1, 4:7]
            pb[m] = ((b[m] - ttc[None, :]) @ Rtc - ext_t[l][None, :]) @ Rl
    scan = np.zeros((sel.size, 12), np.float32)
    scan[:, :3] = pb.astype(np.float32)
    scan[:, 3] = 1.0
    scan[:, 4] = (rng.integers(0, n_table, sel.size) + rng.uniform(0, 0.999, sel.size)).astype(np.float32)
    scan[:, 8] = lid.astype(np.float32)
    return scan


@pytest.mark.gpu
@pytest.mark.parametrize("style", ["call_by_call", "pipelined"])
def test_mapping_loop_over_a_sequence_of_scans(orc, capi, scenes, style):
    """style = pipelined: the loop the way bench.py's `pipelined_turn` drives it - the NEXT scan staged ahead as packed 20-byte
    records from page-locked memory while map_incremental runs (malio_scan_stage), malio_scan_set_packed, list maintenance on
    its own stream, cached neighbours kept where a certificate allows - checked against the ORACLE step by step like the plain
    loop (not against another run of the library)."""
    rng = np.random.default_rng(2024)
    L, N, K, n_table = 2, 8000, 8, 10
    base = scenes.make_scene(N=2000, Nmap=20000, L=L, seed=5)      # params, tables, temporal comp, extrinsics
    prm = dict(base["params"])
    tables, tc = base["tables"], base["temporal_comp"]
    st = scenes.unpack_state(base["state_gt"], L)
    ext_q, ext_t = st["offR"].copy(), st["offT"].copy()
    world = _world(scenes, rng)
    ds = float(prm["filter_size_map"])

    def pose(k):
        pos = scenes.SURFACE_SHIFT + np.array([3.3 + 2.5 * k, -2.1 + 0.3 * k, 1.8])
        rot = scenes.q_norm(scenes.q_mul(scenes.q_from_rotvec([0, 0, np.deg2rad(31.0 + 2.0 * k)]),
                                         scenes.q_from_rotvec([0.01, -0.02, 0])))
        return pos, rot

    p0, _ = pose(0)
    d0 = np.linalg.norm(world[:, :3] - p0[None, :].astype(np.float32), axis=1)
    map0 = world[d0 < 22.0].copy()                                  # the map knows 22 m, the LiDARs see 35 m
    eng = capi.Engine(prm)
    pipelined = style == "pipelined"
    if pipelined:
        eng.set_option("search_skip", 1)
        pins = [capi.PinnedArray((N, 5), np.float32) for _ in range(2)]
    eng.map_build(map0)
    port = orc.VoxMap(ds)
    port.build(map0)
    o = orc.Oracle(prm, threads=4, use_ref=True)
    o.map_build(map0)
    P = scenes.init_P(L)
    post = None
    sizes = []
    scans = [_scan_from(scenes, rng, world, *pose(k), ext_q, ext_t, tc, N, L, 35.0, n_table) for k in range(K)]
    assert all(sc_.shape[0] == N for sc_ in scans)
    if pipelined:
        pins[0].array[:] = capi.Engine.pack_scan(scans[0])
        eng.scan_stage(pins[0].array, True)
    for k in range(K):
        pos, rot = pose(k)
        scan = scans[k]
        # prediction: ground truth + a bounded error (stands in for the IMU propagation of laserMapping.cpp:987)
        dpos = rng.normal(size=3)
        dpos *= 0.06 / np.linalg.norm(dpos)
        drot = rng.normal(size=3)
        drot *= np.deg2rad(0.3) / np.linalg.norm(drot)
        if post is None:
            offR, offT = ext_q, ext_t
        else:
            s = scenes.unpack_state(post, L)
            offR, offT = s["offR"], s["offT"]                       # the filter keeps refining the extrinsics
        prior = scenes.pack_state(pos + dpos, scenes.q_norm(scenes.q_mul(rot, scenes.q_from_rotvec(drot))), offR, offT)
        if pipelined:
            eng.scan_set_packed(pins[k % 2].array, tables, tc)
        else:
            eng.scan_set(scan, tables, tc)
        o.scan_set(scan, tables, tc)
        u, v = eng.update_iterated(prior, P), o.update_iterated(prior, P)
        if pipelined and k + 1 < K:   # the next scan travels while this one's map_incremental runs
            eng.scan_upload_wait()
            pins[(k + 1) % 2].array[:] = capi.Engine.pack_scan(scans[k + 1])
            eng.scan_stage(pins[(k + 1) % 2].array, True)
        assert (u["passes"], u["searches"], u["M"]) == (v["passes"], v["searches"], v["M"]), "scan %d" % k
        assert np.abs(u["state"] - v["state"]).max() < 1e-8, "scan %d" % k
        conftest_assert_P(u["P"], v["P"])
        got = scenes.unpack_state(u["state"], L)
        assert np.linalg.norm(got["pos"] - pos) < 0.06   # sanity only: both sides agree to 1e-8 above
        post = v["state"]
        # map_incremental at the posterior (both sides at the oracle's posterior)
        wny = np.full(scan.shape[0], 0.001 if k == 0 else 0.0, np.float32)   # laserMapping.cpp:1004 vs PCL default
        A, B = o.map_incremental(post, True, wny)
        na, nn, ret = eng.map_incremental(post, True, wny)
        assert (na, nn) == (A.shape[0], B.shape[0]), "scan %d" % k
        assert port.add(A, True) == ret
        port.add(B, False)
        if k % 3 == 2:   # lasermap_fov_segment: drop what lies far behind
            box = np.array([[-1e4, -1e4, -1e4, pos[0] - 18.0, 1e4, 1e4]], np.float32)
            assert eng.map_delete_boxes(box) == port.delete_boxes(box)
        assert eng.map_size() == port.size()
        m_port = port.flatten()
        a4 = np.ascontiguousarray(eng.map_get()[:, [0, 1, 2, 5]])
        b4 = np.ascontiguousarray(m_port[:, [0, 1, 2, 5]])
        np.testing.assert_array_equal(a4[np.lexsort(a4.T[::-1])], b4[np.lexsort(b4.T[::-1])])
        o.map_build(m_port)                                          # the oracle's tree follows the same set
        sizes.append(port.size())
        P = v["P"] + 1e-5 * np.eye(P.shape[0])
    assert sizes[-1] != sizes[0]
    dbg = eng.debug_counters()
    assert dbg["inplace"] >= K, dbg                                  # the lists followed the map without rebuilds ...
    assert dbg["rebuilds"] <= 3, dbg                                 # ... except when a batch did not fit


def conftest_assert_P(P, Q):
    from conftest import assert_P_close
    assert_P_close(P, Q, rel=5e-3)
