"""a13/a14: SE(3) B-spline + per-point undistortion. CPU: the product's host spline (malio_spline_feed /
malio_spline_get_pose, pure host code in libmalio_hip.so) against the oracle restatement of BsplineSE3.cpp.
GPU: malio_undistort against the oracle's restatement of IMU_Processing.hpp:452-508."""
import numpy as np
import pytest


def make_traj(scenes, rng, t0=1671631987.6, dur=0.32, rate=200.0):
    ts = t0 + np.arange(0, dur, 1.0 / rate)
    out = []
    w = np.array([0.3, -0.2, 1.1])   # rad/s
    v = np.array([8.0, 0.5, -0.2])   # m/s
    for t in ts:
        dt = t - t0
        q = scenes.q_from_rotvec(w * dt + 0.02 * np.sin(7 * dt) * np.array([1, 0.5, 0.2]))
        p = v * dt + 0.05 * np.array([np.sin(5 * dt), np.cos(3 * dt) - 1, dt * dt])
        out.append([t, *p, *q])
    return np.array(out)


def make_case(scenes, seed, n=20000):
    rng = np.random.default_rng(seed)
    traj = make_traj(scenes, rng)
    t0 = traj[0, 0]
    beg, end = t0 + 0.05, t0 + 0.15
    curv = np.sort(rng.uniform(0, (end - beg) * 1000.0, n)).astype(np.float32)  # ms, ascending (time_sort)
    pts = np.zeros((n, 12), np.float32)
    pts[:, :3] = rng.uniform(-60, 60, (n, 3))
    pts[:, 3] = 1
    pts[:, 8] = rng.uniform(0, 255, n)  # reflectivity before undistortion
    pts[:, 9] = curv
    imu_t = traj[::2, 0].copy()  # 100 Hz IMU stamps
    A = rng.normal(size=(len(imu_t), 6, 6)) * 1e-3
    imu_c = np.einsum("kij,klj->kil", A, A) + 1e-6 * np.eye(6)
    ext = scenes.make_pose(scenes.q_norm([0.01, -0.02, 0.7, 0.71]), [0.2, -0.1, 0.05], 1e-6 * np.eye(6))
    return dict(traj=traj, beg=beg, end=end, pts=pts, imu_t=imu_t, imu_c=imu_c.reshape(len(imu_t), 36), ext=ext)


def cov_pointer0(imu_t, end):
    cp = len(imu_t) - 1  # IMU_Processing.hpp:453-467
    while True:
        if imu_t[cp] > end:
            cp -= 1
        else:
            cp += 1
            break
    return cp


def test_host_spline_matches_oracle(capi, orc, scenes):
    rng = np.random.default_rng(1)
    traj = make_traj(scenes, rng)
    sp = orc.Spline(traj)
    ot, oT = sp.control()
    gt, gT = capi.spline_feed(traj)
    assert len(gt) == len(ot) and np.array_equal(gt, ot)
    assert np.allclose(gT, oT, rtol=0, atol=1e-12)
    for ts in np.concatenate([traj[0, 0] + rng.uniform(-0.02, 0.34, 300), ot[:5], ot[-5:]]):
        ok_o, q_o, p_o = sp.get_pose(ts)
        ok_g, q_g, p_g = capi.spline_get_pose(gt, gT, ts)
        assert ok_o == ok_g
        if ok_o:
            assert np.allclose(p_g, p_o, rtol=0, atol=1e-11) and np.allclose(q_g, q_o, rtol=0, atol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n", [(3, 20000), (4, 257), (5, 200000)])
def test_undistort_matches_oracle(capi, orc, scenes, seed, n):
    cs = make_case(scenes, seed, n)
    sp = orc.Spline(cs["traj"])
    ok, lq, lt = sp.get_pose(cs["end"])
    assert ok
    ltf = scenes.make_pose(lq, lt, 1e-6 * np.eye(6))
    want, unc = sp.undistort(cs["pts"], cs["beg"], cs["end"], cs["imu_t"], cs["imu_c"], cs["ext"], ltf)
    eng = capi.Engine(scenes.DEFAULT_PARAMS, device=0)
    kt, kT = capi.spline_feed(cs["traj"])
    got, entries = eng.undistort(cs["pts"], cs["beg"], kt, kT, cs["ext"][0:4], cs["ext"][4:7], lq, lt, cs["imu_t"],
                                 cov_pointer0(cs["imu_t"], cs["end"]))
    # first point untouched (loop bounds :475-476); uncertainty-interval index exact; coordinates within 1 float ulp
    assert np.array_equal(got[0], cs["pts"][0]) and np.array_equal(want[0], cs["pts"][0])
    assert np.array_equal(got[:, 8], want[:, 8])
    assert np.array_equal(got[:, 9], want[:, 9])
    d = np.abs(got[:, :3].astype(np.float64) - want[:, :3])
    assert d.max() <= 1.5e-5  # 1 ulp of float32 at 100 m is 7.6e-6
    assert np.mean(got[:, :3] == want[:, :3]) > 0.999
    assert len(entries) == len(unc) > 3
    # the entry points are where the index steps (descending order of processing)
    idx = want[:, 8]
    steps = [i for i in range(n - 1, 0, -1) if (i == n - 1 and idx[i] == 0) or (i < n - 1 and idx[i] > idx[i + 1])]
    assert list(entries) == steps
