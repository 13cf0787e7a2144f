"""Developer aid: wall time of the resident front end of one LiDAR (raw points -> scan installed), 200 k raw points."""
import sys, time
import numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import __graft_entry__ as ge
ge.load_package()
from malio_amd import capi, scenes
sc = scenes.make_scene(cfg=1)  # one LiDAR
e = capi.Engine(sc["params"]); e.map_build(sc["map"])
rng = np.random.default_rng(5)
n = 200_000
t0 = 1671631987.6
ts = t0 + np.arange(0, 0.32, 1.0 / 200.0)
traj = np.array([[t, *(np.array([8.0, 0.5, -0.2]) * (t - t0)), *scenes.q_from_rotvec(np.array([0.3, -0.2, 1.1]) * (t - t0))] for t in ts])
beg, end = t0 + 0.05, t0 + 0.15
pts = np.zeros((n, 12), np.float32)
pts[:, :3] = rng.uniform(-60, 60, (n, 3))
pts[:, 9] = np.sort(rng.uniform(0, (end - beg) * 1000.0, n)).astype(np.float32)
kt, kT = capi.spline_feed(traj)
_, q_end, p_end = capi.spline_get_pose(kt, kT, end)
imu_t = traj[::2, 0].copy()
cp = int(np.searchsorted(imu_t, end, side="right"))
ext_q, ext_t = scenes.q_norm([0.01, -0.02, 0.7, 0.71]), np.array([0.2, -0.1, 0.05])
pin_in, pin_out = capi.PinnedArray((n, 12), np.float32), capi.PinnedArray((n, 12), np.float32)
pin_in.array[:] = pts
for k in range(16):
    pinned = (k // 2) % 2 == 1
    src = pin_in.array if pinned else pts
    t = time.perf_counter()
    e.undistort_resident(0, src, beg, kt, kT, ext_q, ext_t, q_end, p_end, imu_t, cp)
    t1 = time.perf_counter()
    body = e.scan_set_resident(0.5, sc["tables"], sc["temporal_comp"], want_body=(k % 2 == 0), out=pin_out.array if pinned else None)
    t2 = time.perf_counter()
    print("turn %d (%s host buffers): undistort_resident %.2f ms  scan_set_resident(want_body=%d) %.2f ms  -> %d points" % (
        k, "pinned" if pinned else "pageable", (t1 - t) * 1e3, k % 2 == 0, (t2 - t1) * 1e3, e.N))
