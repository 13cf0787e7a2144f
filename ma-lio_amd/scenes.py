"""Seeded synthetic workloads for the measurement-update hot path.

There is no dataset access here, so every parity test and bench line runs on these scenes:
a ground plane plus building walls (planes x = 40 j, y = 60 j, 0..20 m high) or a straight
tunnel (config 5), a 0.5 m-voxel map, and an N-point fused multi-LiDAR scan expressed in the
per-LiDAR frames through the City.yaml / UrbanNav.yaml extrinsics
(/root/reference/MA_LIO/config/City.yaml:24-29, UrbanNav.yaml:24-27).

Layouts produced (see include/malio.h):
  points : [n,12] float32 = pcl::PointXYZINormal (x y z _ normal_x normal_y normal_z _ intensity curvature _ _)
  pose   : [59] float64  = q(x,y,z,w) t(3) T(4x4 row-major) cov(6x6 row-major)
  state  : [19+7L] float64 = pos rot(x,y,z,w) offset_R[L] offset_T[L] vel bg ba grav
"""
import numpy as np

CITY_EXT_T = np.array([[0.215, 0, 0.018], [-1.2574, 0.413, 0.0324], [-1.306, -0.361, 0.042]])
# yaml order is (w,x,y,z) (laserMapping.cpp:843); stored here as (x,y,z,w)
CITY_EXT_Q_WXYZ = np.array([[1, 0, 0, 0], [0.6965018, -0.0037329, -0.0038405, 0.717535],
                            [0.0074645, 0.0000044, -0.0005919, -0.999972]])
URBAN_EXT_T = np.array([[0, 0, 0.28], [0.3237, -0.0012, 0.0791]])
URBAN_EXT_Q_WXYZ = np.array([[1, 0, 0, 0], [0.8849, 0.0027, 0.4654, -0.0182]])

SURFACE_SHIFT = np.array([17.0, 23.0, -1.8])

DEFAULT_PARAMS = dict(  # City.yaml:41-49, mapping_city.launch:9-15
    lid_num=3, max_iteration=3, extrinsic_est_en=1, plane_th=0.4, cov_threshold=0.5, range_min=0.0, range_max=1.0,
    point_cov_max=0.00125, point_cov_min=0.00075, plane_cov_max=1.0, plane_cov_min=0.8, localize_cov_max=2.0,
    localize_cov_min=0.3, localize_thresh_max=0.7, localize_thresh_min=0.2, filter_size_map=0.5, limit=0.0)

# BASELINE.json configs (index = config number - 1); seeds 20230625 + config index
CONFIGS = {
    1: dict(name="velodyne16_10k_50k", N=10_000, Nmap=50_000, L=1, max_iteration=3, kind="city", map_unc=False),
    2: dict(name="city3_100k_1M", N=100_000, Nmap=1_000_000, L=3, max_iteration=3, kind="city", map_unc=False),
    3: dict(name="urban2_60k_500k_unc", N=60_000, Nmap=500_000, L=2, max_iteration=3, kind="city", map_unc=True),
    4: dict(name="synth3_200k_8M", N=200_000, Nmap=8_000_000, L=3, max_iteration=3, kind="city", map_unc=False),
    # "10 IESKF iterations": with the reference's limit of 1e-3 the loop stops after 4 passes on this scene, so the
    # config tightens esekf's `limit` (esekfom.hpp:160-163,649-657) until it never fires: passes -1..8 all run, the
    # forced search of :660-663 at i == maximum_iter - 2 included
    5: dict(name="tunnel3_100k_1M_10it", N=100_000, Nmap=1_000_000, L=3, max_iteration=9, kind="tunnel",
            map_unc=False, limit=1e-30),
}


# ------------------------------------------------------------------------------------------------
# quaternion helpers, (x,y,z,w) Hamilton
def q_mul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by + ay * bw + az * bx - ax * bz,
                     aw * bz + az * bw + ax * by - ay * bx, aw * bw - ax * bx - ay * by - az * bz])


def q_conj(q):
    return np.array([-q[0], -q[1], -q[2], q[3]])


def q_to_R(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def q_from_rotvec(v):
    v = np.asarray(v, float)
    th = np.linalg.norm(v)
    if th < 1e-12:
        return np.array([0.5 * v[0], 0.5 * v[1], 0.5 * v[2], 1.0])
    s = np.sin(th / 2) / th
    return np.array([v[0] * s, v[1] * s, v[2] * s, np.cos(th / 2)])


def q_norm(q):
    q = np.asarray(q, float)
    return q / np.linalg.norm(q)


def make_pose(q, t, cov=None):
    """Pack a common_lib.h:57-63 Pose as 59 doubles."""
    p = np.zeros(59)
    p[0:4] = q
    p[4:7] = t
    T = np.eye(4)
    T[:3, :3] = q_to_R(q)
    T[:3, 3] = t
    p[7:23] = T.reshape(-1)
    if cov is not None:
        p[23:59] = np.asarray(cov, float).reshape(-1)
    return p


def pack_state(pos, rot, offR, offT, vel=(0, 0, 0), bg=(0, 0, 0), ba=(0, 0, 0), grav=(0, 0, -9.809)):
    return np.concatenate([np.asarray(pos, float), np.asarray(rot, float), np.asarray(offR, float).reshape(-1),
                           np.asarray(offT, float).reshape(-1), np.asarray(vel, float), np.asarray(bg, float),
                           np.asarray(ba, float), np.asarray(grav, float)])


def unpack_state(s, L):
    s = np.asarray(s, float)
    o = 0
    out = {}
    out["pos"] = s[o:o + 3]; o += 3
    out["rot"] = s[o:o + 4]; o += 4
    out["offR"] = s[o:o + 4 * L].reshape(L, 4); o += 4 * L
    out["offT"] = s[o:o + 3 * L].reshape(L, 3); o += 3 * L
    out["vel"] = s[o:o + 3]; o += 3
    out["bg"] = s[o:o + 3]; o += 3
    out["ba"] = s[o:o + 3]; o += 3
    out["grav"] = s[o:o + 3]
    return out


def init_P(L):
    """Initial covariance pattern of IMU_Processing.hpp:184-199."""
    n = 17 + 6 * L
    P = np.eye(n)
    for i in range(6, n):
        if i < n - 8:
            P[i, i] = 0.000001
        elif i < n - 5:
            P[i, i] = 0.0001
        elif i < n - 2:
            P[i, i] = 0.001
        else:
            P[i, i] = 0.00001
    return P


# ------------------------------------------------------------------------------------------------
def _surface_voxels(kind, half_w, rng, origin):
    """One point per occupied 0.5 m surface voxel inside |x|,|y| <= half_w (about `origin`).
    Returns [n,3] float64 world points (in-plane uniform inside the voxel, N(0,0.02) off-plane)."""
    v = 0.5
    pts = []
    if kind == "city":
        nx = int(np.floor(half_w / v))
        gx = (np.arange(-nx, nx) + 0.0) * v
        X, Y = np.meshgrid(gx, gx, indexing="ij")
        n = X.size
        ground = np.stack([X.ravel() + rng.uniform(0, v, n), Y.ravel() + rng.uniform(0, v, n),
                           rng.normal(0, 0.02, n)], 1)
        pts.append(ground)
        hz = (np.arange(0, 40)) * v  # 0..20 m
        for j in range(-int(half_w // 40), int(half_w // 40) + 1):  # walls x = 40 j
            Yw, Zw = np.meshgrid(gx, hz, indexing="ij")
            m = Yw.size
            pts.append(np.stack([40.0 * j + rng.normal(0, 0.02, m), Yw.ravel() + rng.uniform(0, v, m),
                                 Zw.ravel() + rng.uniform(0, v, m)], 1))
        for j in range(-int(half_w // 60), int(half_w // 60) + 1):  # walls y = 60 j
            Xw, Zw = np.meshgrid(gx, hz, indexing="ij")
            m = Xw.size
            pts.append(np.stack([Xw.ravel() + rng.uniform(0, v, m), 60.0 * j + rng.normal(0, 0.02, m),
                                 Zw.ravel() + rng.uniform(0, v, m)], 1))
    elif kind == "plain":  # ground plane only: two unobservable translations (localization-weight min branch)
        nx = int(np.floor(half_w / v))
        gx = (np.arange(-nx, nx) + 0.0) * v
        X, Y = np.meshgrid(gx, gx, indexing="ij")
        n = X.size
        pts.append(np.stack([X.ravel() + rng.uniform(0, v, n), Y.ravel() + rng.uniform(0, v, n),
                             rng.normal(0, 0.02, n)], 1))
    else:  # tunnel along +x: walls y = +-5, floor z = 0, ceiling z = 6, no cross features
        nx = int(np.floor(half_w / v))
        gx = (np.arange(-nx, nx) + 0.0) * v
        gy = np.arange(-10, 10) * v
        gz = np.arange(0, 12) * v
        for zc in (0.0, 6.0):
            X, Y = np.meshgrid(gx, gy, indexing="ij")
            m = X.size
            pts.append(np.stack([X.ravel() + rng.uniform(0, v, m), Y.ravel() + rng.uniform(0, v, m),
                                 zc + rng.normal(0, 0.02, m)], 1))
        for yc in (-5.0, 5.0):
            X, Z = np.meshgrid(gx, gz, indexing="ij")
            m = X.size
            pts.append(np.stack([X.ravel() + rng.uniform(0, v, m), yc + rng.normal(0, 0.02, m),
                                 Z.ravel() + rng.uniform(0, v, m)], 1))
    p = np.concatenate(pts, 0)
    return p + np.asarray(origin, float)[None, :]


def _count_for(kind, half_w):
    v = 0.5
    nx = 2 * int(np.floor(half_w / v))
    if kind == "city":
        return nx * nx + (2 * int(half_w // 40) + 1) * nx * 40 + (2 * int(half_w // 60) + 1) * nx * 40
    if kind == "plain":
        return nx * nx
    return 2 * nx * 20 + 2 * nx * 12


def make_map(kind, Nmap, rng, origin=(0, 0, 0), map_unc=False):
    """Map grown (square half-width) until it holds >= Nmap voxels, then cut to exactly Nmap."""
    lo, hi = 5.0, 20000.0
    for _ in range(60):
        mid = 0.5 * (lo + hi)
        if _count_for(kind, mid) >= Nmap:
            hi = mid
        else:
            lo = mid
    half_w = hi
    p = _surface_voxels(kind, half_w, rng, origin)
    if p.shape[0] > Nmap:
        keep = rng.permutation(p.shape[0])[:Nmap]
        keep.sort()
        p = p[keep]
    pts = np.zeros((p.shape[0], 12), np.float32)
    pts[:, 0:3] = p.astype(np.float32)
    pts[:, 3] = 1.0
    pts[:, 5] = rng.uniform(0, 0.002, p.shape[0]).astype(np.float32) if map_unc else np.float32(0.001)
    return pts, half_w


def make_scene(cfg=None, seed=None, N=None, Nmap=None, L=None, kind="city", map_unc=False, origin=(0, 0, 0),
               max_iteration=3, extrinsic_est_en=1, n_table=10, prior_dpos=0.10, prior_drot_deg=0.5,
               det_range=100.0, scan_seed=None, limit=0.0):
    """Build one synthetic scan-vs-map problem. `cfg` selects a BASELINE.json config (1..5)."""
    if cfg is not None:
        c = CONFIGS[cfg]
        N, Nmap, L, kind, map_unc = c["N"], c["Nmap"], c["L"], c["kind"], c["map_unc"]
        max_iteration = c["max_iteration"]
        limit = c.get("limit", limit)
        if kind == "tunnel":
            det_range = 500.0  # a 10 m x 6 m tunnel only offers 128 voxels per metre of length
        if seed is None:
            seed = 20230625 + cfg
    if seed is None:
        seed = 20230625
    rng = np.random.default_rng(seed)
    # The reference fits planes as a x + b y + c z = -1 (common_lib.h:156-174), which is singular for planes
    # through the world origin; real maps start at the first IMU pose (ground ~1.8 m below, no wall through
    # the origin), so the synthetic surfaces are shifted accordingly. `origin` adds on top (0 m / 1000 m).
    origin = np.asarray(origin, float) + SURFACE_SHIFT
    map_pts, half_w = make_map(kind, Nmap, rng, origin, map_unc)

    # ground-truth pose
    pos_gt = origin + np.array([3.3, -2.1, 1.8])
    rot_gt = q_norm(q_mul(q_from_rotvec([0, 0, np.deg2rad(31.0)]), q_from_rotvec([0.01, -0.02, 0])))
    R_gt = q_to_R(rot_gt)
    if L == 3:
        ext_t, ext_q = CITY_EXT_T.copy(), CITY_EXT_Q_WXYZ[:, [1, 2, 3, 0]].copy()
        split = [0.60, 0.25, 0.15]
    elif L == 2:
        ext_t, ext_q = URBAN_EXT_T.copy(), URBAN_EXT_Q_WXYZ[:, [1, 2, 3, 0]].copy()
        split = [0.65, 0.35]
    else:
        ext_t, ext_q = np.zeros((L, 3)), np.tile(np.array([0, 0, 0, 1.0]), (L, 1))
        split = [1.0 / L] * L
    ext_q = np.array([q_norm(q) for q in ext_q])

    # temporal compensation (L-1 poses): <= 5 cm, <= 0.2 deg
    tc = []
    for l in range(L - 1):
        ax = rng.normal(size=3)
        ax /= np.linalg.norm(ax)
        q = q_from_rotvec(ax * np.deg2rad(rng.uniform(0, 0.2)))
        t = rng.uniform(-0.05, 0.05, 3) / np.sqrt(3)
        cov = np.diag([1e-6] * 3 + [1e-7] * 3)
        tc.append(make_pose(q, t, cov))
    tc = np.array(tc).reshape(-1, 59)

    # pose_unc tables: covariance diag growing linearly 1e-6 -> 1e-4 with idx, small transforms
    tables = []
    for l in range(L):
        tab = []
        for k in range(n_table):
            s = 1e-6 + (1e-4 - 1e-6) * k / max(n_table - 1, 1)
            A = rng.normal(size=(6, 6)) * 0.05
            cov = s * (np.eye(6) + A @ A.T)  # SPD, mildly non-diagonal so the cross terms are exercised
            ax = rng.normal(size=3)
            ax /= np.linalg.norm(ax)
            q = q_from_rotvec(ax * np.deg2rad(0.1) * (k + 1) / n_table)
            t = rng.normal(size=3) * 0.01 * (k + 1) / n_table
            tab.append(make_pose(q, t, cov))
        tables.append(np.array(tab))

    # scan: N surface points within det_range of the sensor, one per 0.5 m voxel, noise N(0,0.02)
    if scan_seed is not None:  # a different scan over the same map / tables (multi-GPU weak scaling shards)
        rng = np.random.default_rng(scan_seed)
    rel = map_pts[:, 0:3].astype(np.float64) - pos_gt[None, :]
    near = np.nonzero(np.einsum("ij,ij->i", rel, rel) < det_range * det_range)[0]
    if near.size < N:
        raise ValueError(f"only {near.size} surface voxels within range for N={N}")
    sel = rng.permutation(near)[:N]
    v = 0.5
    base = map_pts[sel, 0:3].astype(np.float64)
    # re-draw inside the same voxel so scan points are not copies of map points
    pw = (np.floor((base - origin) / v) * v + origin) + rng.uniform(0, v, (N, 3))
    # snap the off-plane coordinate back to the map point's (surface) value plus sensor noise
    plane_axis = np.argmin(_offplane_hint(kind, base - origin), axis=1)
    idx = np.arange(N)
    pw[idx, plane_axis] = base[idx, plane_axis] + rng.normal(0, 0.02, N)
    lid = rng.choice(L, size=N, p=np.array(split) / np.sum(split))
    b = (pw - pos_gt[None, :]) @ R_gt  # R_gt^T (p_w - pos)
    pb = np.zeros((N, 3))
    for l in range(L):
        m = lid == l
        if not m.any():
            continue
        Rl = q_to_R(ext_q[l])
        if l == 0:
            pb[m] = (b[m] - ext_t[0][None, :]) @ Rl
        else:
            Rtc = q_to_R(tc[l - 1, 0:4])
            ttc = tc[l - 1, 4:7]
            pb[m] = ((b[m] - ttc[None, :]) @ Rtc - ext_t[l][None, :]) @ Rl
    scan = np.zeros((N, 12), np.float32)
    scan[:, 0:3] = pb.astype(np.float32)
    scan[:, 3] = 1.0
    # normal_x = mean uncertainty-interval index after the voxel filter (a float; truncated on use)
    scan[:, 4] = (rng.integers(0, n_table, N) + rng.uniform(0, 0.999, N)).astype(np.float32)
    scan[:, 8] = lid.astype(np.float32)
    scan[:, 9] = rng.uniform(0, 100.0, N).astype(np.float32)

    # prior state = ground truth boxplus delta
    dpos = rng.normal(size=3)
    dpos *= prior_dpos / np.linalg.norm(dpos)
    drot = rng.normal(size=3)
    drot *= np.deg2rad(prior_drot_deg) / np.linalg.norm(drot)
    rot0 = q_norm(q_mul(rot_gt, q_from_rotvec(drot)))
    state_gt = pack_state(pos_gt, rot_gt, ext_q, ext_t)
    state0 = pack_state(pos_gt + dpos, rot0, ext_q, ext_t)
    params = dict(DEFAULT_PARAMS)
    params.update(lid_num=L, max_iteration=max_iteration, extrinsic_est_en=extrinsic_est_en, limit=limit)
    return dict(params=params, map=map_pts, scan=scan, tables=tables, temporal_comp=tc, state0=state0,
                state_gt=state_gt, P0=init_P(L), L=L, N=N, Nmap=map_pts.shape[0], seed=seed, kind=kind,
                half_w=half_w)


def _offplane_hint(kind, rel):
    """Per-point score per axis; the surface normal axis of the voxel a map point came from is the one
    whose coordinate sits (almost) on a generating plane."""
    d = np.full(rel.shape, 10.0)
    if kind == "city":
        d[:, 2] = np.abs(rel[:, 2])  # ground z = 0
        d[:, 0] = np.abs(rel[:, 0] - 40.0 * np.round(rel[:, 0] / 40.0))
        d[:, 1] = np.abs(rel[:, 1] - 60.0 * np.round(rel[:, 1] / 60.0))
    elif kind == "plain":
        d[:, 2] = np.abs(rel[:, 2])
    else:
        d[:, 2] = np.minimum(np.abs(rel[:, 2]), np.abs(rel[:, 2] - 6.0))
        d[:, 1] = np.minimum(np.abs(rel[:, 1] - 5.0), np.abs(rel[:, 1] + 5.0))
    return d
