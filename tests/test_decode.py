"""Sensor decode: City-dataset records -> pl_surf. CPU: the restated handlers against an
independent NumPy formulation. GPU: malio_decode_livox / malio_decode_ouster / malio_decode_velodyne against the
restatement, bit for bit."""
import numpy as np
import pytest


def livox_records(rng, n, tele=False):
    """n synthetic 19-byte records in acquisition order (offset_time ascending), with the cases the handler branches
    on: invalid tags/lines, repeated points, points inside the blind sphere, offsets beyond 100 ms."""
    rec = np.zeros(n, dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("r", "u1"), ("tag", "u1"), ("line", "u1"),
                             ("t", "<u4")])
    rec["x"], rec["y"], rec["z"] = rng.uniform(-60, 60, n), rng.uniform(-60, 60, n), rng.uniform(-3, 20, n)
    rec["r"] = rng.integers(0, 256, n)
    rec["tag"] = rng.choice([0x00, 0x10, 0x20, 0x30, 0x11, 0x05], n, p=[0.45, 0.35, 0.05, 0.05, 0.05, 0.05])
    rec["line"] = rng.integers(0, 8 if not tele else 2, n)
    rec["t"] = np.sort(rng.integers(0, 120_000_000, n)).astype(np.uint32)      # ns; > 1e8 -> curvature > 100 ms
    k = n // 20
    idx = rng.integers(1, n, k)
    for f in ("x", "y", "z"):
        rec[f][idx] = rec[f][idx - 1]                                            # exact repeats of the previous point
    near = rng.integers(0, n, k)
    rec["x"][near], rec["y"][near], rec["z"][near] = rng.uniform(-0.3, 0.3, k), rng.uniform(-0.3, 0.3, k), rng.uniform(-0.3, 0.3, k)
    b = rec.tobytes()
    assert len(b) == 19 * n
    return np.frombuffer(b, np.uint8), rec


def ouster_records(rng, n):
    rec = np.zeros(n, dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("i", "<f4"), ("ring", "<u2"), ("t", "<u4")])
    rec["x"], rec["y"], rec["z"] = rng.uniform(-80, 80, n), rng.uniform(-80, 80, n), rng.uniform(-3, 25, n)
    rec["i"] = rng.uniform(0, 4000, n)
    rec["ring"] = rng.integers(0, 128, n)
    rec["t"] = np.sort(rng.integers(0, 100_000_000, n)).astype(np.uint32)
    near = rng.integers(0, n, n // 10)
    rec["x"][near], rec["y"][near], rec["z"][near] = 0.1, -0.2, 0.05
    b = rec.tobytes()
    assert len(b) == 22 * n
    return np.frombuffer(b, np.uint8), rec


def numpy_avia(rec, n_scans, pfn, blind, eof_point):
    if eof_point:
        rec = np.concatenate([rec, np.zeros(1, rec.dtype)])
    n = rec.shape[0]
    valid = (rec["line"] < n_scans) & (((rec["tag"] & 0x30) == 0x10) | ((rec["tag"] & 0x30) == 0x00))
    valid[0] = False
    vnum = np.cumsum(valid)
    looked = valid & (vnum % pfn == 0)
    curv = rec["t"].astype(np.float32) / np.float32(1000000)
    full = np.zeros((n, 3), np.float32)
    full[looked] = np.stack([rec["x"], rec["y"], rec["z"]], 1)[looked]
    prev = np.vstack([np.zeros((1, 3), np.float32), full[:-1]])
    xyz = np.stack([rec["x"], rec["y"], rec["z"]], 1)
    d = np.abs(xyz - prev) > 1e-7
    far = (xyz[:, 0] * xyz[:, 0] + xyz[:, 1] * xyz[:, 1] + xyz[:, 2] * xyz[:, 2]).astype(np.float64) > blind * blind
    keep = looked & ~(curv > 100) & (d[:, 0] | d[:, 1] | (d[:, 2] & far))
    tm = curv[looked & ~(curv > 100)]
    return xyz[keep], rec["r"][keep].astype(np.float32), curv[keep], (float(tm.max()) if tm.size else -9999.0)


@pytest.mark.parametrize("seed,n,pfn,eof", [(1, 24000, 3, False), (2, 5000, 1, True), (3, 100, 4, True), (4, 2, 1, False)])
def test_oracle_livox_decode_matches_numpy(orc, seed, n, pfn, eof):
    rng = np.random.default_rng(seed)
    b, rec = livox_records(rng, n)
    out, mt = orc.decode_livox(b, 6, pfn, 1.0, eof)
    xyz, refl, curv, tmax = numpy_avia(rec, 6, pfn, 1.0, eof)
    assert out.shape[0] == xyz.shape[0]
    np.testing.assert_array_equal(out[:, :3], xyz)
    np.testing.assert_array_equal(out[:, 8], refl)
    np.testing.assert_array_equal(out[:, 9], curv)
    assert mt == tmax and (out[:, 4:8] == 0).all()


@pytest.mark.parametrize("seed,n,pfn", [(1, 65536, 4), (2, 1000, 1), (3, 7, 3)])
def test_oracle_ouster_decode_matches_numpy(orc, seed, n, pfn):
    rng = np.random.default_rng(seed)
    b, rec = ouster_records(rng, n)
    out, mt = orc.decode_ouster(b, pfn, 2.0, 1.0e-3)
    idx = np.arange(n)
    x, y, z = rec["x"], rec["y"], rec["z"]
    keep = (idx % pfn == 0) & ~((x * x + y * y + z * z).astype(np.float64) < 4.0)
    curv = rec["t"].astype(np.float32) * np.float32(1.0e-3) * np.float32(1.e-9)
    assert out.shape[0] == keep.sum()
    np.testing.assert_array_equal(out[:, 0], x[keep])
    np.testing.assert_array_equal(out[:, 8], rec["i"][keep])
    np.testing.assert_array_equal(out[:, 9], curv[keep])
    assert mt == (float(curv[keep].max()) if keep.any() else -9999.0)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n,pfn,eof", [(1, 240000, 3, False), (2, 50000, 1, True), (3, 100, 4, True), (4, 2, 1, False),
                                            (5, 0, 2, True), (6, 1, 1, True)])
def test_gpu_livox_decode_equals_oracle(orc, capi, scenes, seed, n, pfn, eof):
    rng = np.random.default_rng(seed)
    b, _ = livox_records(rng, n) if n else (np.zeros(0, np.uint8), None)
    eng = capi.Engine(scenes.make_scene(cfg=1)["params"])
    got, mt_g = eng.decode_livox(b, 6, pfn, 1.0, eof)
    want, mt_o = orc.decode_livox(b, 6, pfn, 1.0, eof)
    np.testing.assert_array_equal(got, want)
    assert mt_g == mt_o


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n,pfn", [(1, 131072, 4), (2, 65536, 1), (3, 7, 3), (4, 0, 2)])
def test_gpu_ouster_decode_equals_oracle(orc, capi, scenes, seed, n, pfn):
    rng = np.random.default_rng(seed)
    b, _ = ouster_records(rng, n) if n else (np.zeros(0, np.uint8), None)
    eng = capi.Engine(scenes.make_scene(cfg=1)["params"])
    got, mt_g = eng.decode_ouster(b, pfn, 2.0, 1.0e-3)
    want, mt_o = orc.decode_ouster(b, pfn, 2.0, 1.0e-3)
    np.testing.assert_array_equal(got, want)
    assert mt_g == mt_o


# ---- Velodyne: PointCloud2 payload bytes through pcl::fromROSMsg + Preprocess::velodyne_handler (preprocess.cpp:148-212) ----
VEL_DRIVER = (22, 0, 4, 8, 12, 18)     # velodyne_pointcloud's PointXYZIRT: x y z intensity (f32) ring (u16 @16) time (f32 @18)
VEL_PCL = (32, 0, 4, 8, 16, 20)        # velodyne_ros::Point as PCL lays it out in memory (xyz + pad, intensity, time, ring)
VEL_NO_TIME = (16, 0, 4, 8, 12, -1)    # a message without a time field (fromROSMsg leaves it 0)


def velodyne_payload(rng, n, layout, end_stamped=False):
    """n points of a 16-ring sweep as PointCloud2 data[] in the given field layout; offsets relative to the start of the
    sweep (>= 0) or to its END (<= 0, some drivers), a tenth of the points inside the blind sphere, a few NaN returns."""
    step, ox, oy, oz, oi, ot = layout
    buf = np.zeros((n, step), np.uint8)
    if n:
        buf[:] = rng.integers(0, 256, (n, step), dtype=np.uint8)  # whatever sits between the fields must not matter
    xyz = np.stack([rng.uniform(-70, 70, n), rng.uniform(-70, 70, n), rng.uniform(-3, 15, n)], 1).astype(np.float32)
    near = rng.integers(0, max(n, 1), n // 10)
    if n:
        xyz[near] = rng.uniform(-0.5, 0.5, (near.size, 3)).astype(np.float32)
    inten = rng.uniform(0, 255, n).astype(np.float32)
    t = np.sort(rng.uniform(0, 0.1, n)).astype(np.float32)
    if end_stamped:
        t = (t - np.float32(0.1)).astype(np.float32)
    if n > 50:
        t[rng.integers(0, n, 3)] = np.float32("nan")

    def put(off, v):
        if off >= 0 and n:
            buf[:, off:off + 4] = v.astype("<f4").view(np.uint8).reshape(n, 4)
    put(ox, xyz[:, 0]), put(oy, xyz[:, 1]), put(oz, xyz[:, 2]), put(oi, inten), put(ot, t)
    return buf.reshape(-1), xyz, (inten if oi >= 0 else np.zeros(n, np.float32)), (t if ot >= 0 else np.zeros(n, np.float32))


VEL_CASES = [(1, 28800, VEL_DRIVER, 3, False), (2, 5000, VEL_PCL, 1, True), (3, 777, VEL_NO_TIME, 4, False),
             (4, 1, VEL_DRIVER, 1, True), (5, 0, VEL_DRIVER, 2, False)]


@pytest.mark.parametrize("seed,n,layout,pfn,end_stamped", VEL_CASES)
def test_oracle_velodyne_decode_matches_numpy(orc, seed, n, layout, pfn, end_stamped):
    rng = np.random.default_rng(seed)
    b, xyz, inten, t = velodyne_payload(rng, n, layout, end_stamped)
    blind, tus = 2.0, 1.0e3  # time_unit_scale for SEC (preprocess.cpp:25-26): curvature in ms
    out, mt = orc.decode_velodyne(b, n, layout, pfn, blind, tus, maximum_time_in=123.0)
    idx = np.arange(n)
    r2 = (xyz[:, 0] * xyz[:, 0] + xyz[:, 1] * xyz[:, 1] + xyz[:, 2] * xyz[:, 2]).astype(np.float64)
    keep = (idx % pfn == 0) & (r2 > blind * blind)
    curv = t * np.float32(tus)
    assert out.shape[0] == keep.sum()
    np.testing.assert_array_equal(out[:, :3], xyz[keep])
    np.testing.assert_array_equal(out[:, 8], inten[keep])
    np.testing.assert_array_equal(out[:, 9], curv[keep])  # (NaN == NaN positionally)
    assert (out[:, 4:8] == 0).all() and (out[:, 10:] == 0).all()
    if n == 0:
        assert mt == 123.0  # :157-158: the handler returned before resetting maximum_time
    else:
        want = -9999.0
        for c in curv[keep]:  # `if (maximum_time < curvature)`: false for NaN
            if want < float(c):
                want = float(c)
        assert mt == want


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n,layout,pfn,end_stamped", VEL_CASES + [(6, 230400, VEL_DRIVER, 2, False)])
def test_gpu_velodyne_decode_equals_oracle(orc, capi, scenes, seed, n, layout, pfn, end_stamped):
    rng = np.random.default_rng(seed)
    b, _, _, _ = velodyne_payload(rng, n, layout, end_stamped)
    eng = capi.Engine(scenes.make_scene(cfg=1)["params"])
    got, mt_g = eng.decode_velodyne(b, n, layout, pfn, 2.0, 1.0e3, maximum_time_in=123.0)
    want, mt_o = orc.decode_velodyne(b, n, layout, pfn, 2.0, 1.0e3, maximum_time_in=123.0)
    np.testing.assert_array_equal(got.view(np.uint32), want.view(np.uint32))  # bit for bit, NaN curvatures included
    assert mt_g == mt_o


@pytest.mark.gpu
def test_velodyne_decode_feeds_config1_chain(orc, capi, scenes):
    """BASELINE config 1 is a Velodyne-16 configuration: its own sensor's decode output goes through the voxel filter and
    comes out as a scan the update accepts (decode -> down-sample -> scan_set -> measure), GPU against the oracle's decode."""
    sc = scenes.make_scene(cfg=1)
    eng = capi.Engine(sc["params"])
    n = sc["N"]
    step, ox, oy, oz, oi, ot = VEL_DRIVER
    buf = np.zeros((n, step), np.uint8)
    for off, v in ((ox, sc["scan"][:, 0]), (oy, sc["scan"][:, 1]), (oz, sc["scan"][:, 2]), (oi, np.full(n, 7.0, np.float32)),
                   (ot, np.linspace(0, 0.1, n).astype(np.float32))):
        buf[:, off:off + 4] = np.ascontiguousarray(v, "<f4").view(np.uint8).reshape(n, 4)
    got, mt = eng.decode_velodyne(buf.reshape(-1), n, VEL_DRIVER, 1, 0.5, 1.0e3)
    want, mt_o = orc.decode_velodyne(buf.reshape(-1), n, VEL_DRIVER, 1, 0.5, 1.0e3)
    np.testing.assert_array_equal(got, want)
    assert mt == mt_o and abs(mt - 100.0) < 1e-3 and got.shape[0] > 0.99 * n
    cloud = got.copy()
    cloud[:, 8] = 0.0        # the mapping loop's field shuffle (laserMapping.cpp:972-976): intensity <- LiDAR slot
    cloud[:, 4] = 0.0        # normal_x <- uncertainty-interval index
    eng.map_build(sc["map"])
    eng.scan_set(cloud, sc["tables"], sc["temporal_comp"])
    g = eng.measure(sc["state0"], True)
    assert g["valid"] and g["M"] > 0.5 * cloud.shape[0]
