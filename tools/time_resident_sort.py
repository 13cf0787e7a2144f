"""Developer aid: does the voxel filter's own output order (sorted by voxel index in the LiDAR frame) make the spatial
scan sort unnecessary on the resident path? Realistic raw cloud: the scene's scan points x3 with jitter."""
import os, sys, time
import numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import __graft_entry__ as ge
ge.load_package()
from malio_amd import capi, scenes
sc = scenes.make_scene(cfg=1, N=100000, Nmap=1000000) if False else scenes.make_scene(seed=77, N=100000, Nmap=1000000, L=1)
L = 1
rng = np.random.default_rng(5)
t0 = 1671631987.6
ts = t0 + np.arange(0, 0.32, 1.0 / 200.0)
w, v = np.array([0.02, -0.01, 0.03]), np.array([0.15, 0.02, -0.01])
traj = np.array([[t, *(v * (t - t0)), *scenes.q_from_rotvec(w * (t - t0))] for t in ts])
kt, kT = capi.spline_feed(traj)
beg, end = t0 + 0.05, t0 + 0.15
_, q_end, p_end = capi.spline_get_pose(kt, kT, end)
imu_t = traj[::2, 0].copy()
cp = int(np.searchsorted(imu_t, end, side="right"))
st = scenes.unpack_state(sc["state_gt"], L)
raw = np.repeat(sc["scan"], 3, axis=0).copy()
raw[:, :3] += rng.normal(0, 0.05, size=(raw.shape[0], 3)).astype(np.float32)
raw[:, 9] = np.sort(rng.uniform(0, (end - beg) * 1000.0, raw.shape[0])).astype(np.float32)
raw[:, 8] = 0
e = capi.Engine(sc["params"]); e.map_build(sc["map"])
for keep in (0, 1, 0, 1):
    e.scan_order(0 if keep else 1)  # 0: resident scans keep the voxel filter's order (default), 1: always sort
    e.undistort_resident(0, raw, beg, kt, kT, st["offR"][0], st["offT"][0], q_end, p_end, imu_t, cp)
    e.scan_set_resident(0.5, sc["tables"], sc["temporal_comp"], want_body=False)
    t = time.perf_counter(); r = e.measure(sc["state0"], True); first = time.perf_counter() - t
    fn, out = e.measure_fn(sc["state0"], True)
    for _ in range(20): fn()
    t = time.perf_counter()
    for _ in range(200): fn()
    dt = (time.perf_counter() - t) / 200
    e.set_profiling(True); per = {}
    for _ in range(30):
        e.measure(sc["state0"], True)
        for n_, ms in e.last_kernel_times(): per.setdefault(n_, []).append(ms)
    e.set_profiling(False)
    print("keep_order=%d N=%d M=%d first pass %.1f us, steady pass %.2f us, %s" % (keep, e.N, r["M"], first * 1e6, dt * 1e6, {k: round(float(np.median(v_)) * 1e3, 1) for k, v_ in per.items()}))
