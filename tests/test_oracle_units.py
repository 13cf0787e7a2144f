"""CPU tests of the oracle itself (test infrastructure): every restated block is cross-checked against
an independent NumPy/SciPy formulation, and the k-NN against the reference's own ikd-Tree compiled
from /root/reference (oracle/_ref) when that library is present."""
import numpy as np
import pytest
from scipy.linalg import expm, logm
from scipy.spatial import cKDTree
from scipy.spatial.transform import Rotation


def pts12(xyz, normal_y=0.001):
    p = np.zeros((len(xyz), 12), np.float32)
    p[:, :3] = xyz
    p[:, 3] = 1
    p[:, 5] = normal_y
    return p


# ---- a3: esti_plane (common_lib.h:144-190) ---------------------------------------------------------
@pytest.mark.parametrize("origin,tol", [((0, 0, 0), 2e-5), ((100, -50, 3), 2e-3), ((1000, 800, 10), 5e-2)])
def test_esti_plane_vs_lstsq(orc, origin, tol):
    rng = np.random.default_rng(3)
    for _ in range(200):
        n = rng.normal(size=3)
        n /= np.linalg.norm(n)
        c = np.asarray(origin, float) + rng.normal(size=3)
        b1 = np.cross(n, rng.normal(size=3))
        b1 /= np.linalg.norm(b1)
        b2 = np.cross(n, b1)
        uv = rng.uniform(-1, 1, (5, 2))
        P = c + uv[:, :1] * b1 + uv[:, 1:] * b2 + rng.normal(0, 0.01, (5, 1)) * n
        P32 = P.astype(np.float32)
        W = rng.uniform(0.0005, 0.002, 5).astype(np.float32)
        near = pts12(P32)
        near[:, 5] = W
        ok, pabcd, pcov = orc.esti_plane(near, 0.4, 0.5)
        # independent solve in double on the float-rounded points: A x = -1
        x, *_ = np.linalg.lstsq(P32.astype(np.float64), -np.ones(5), rcond=None)
        nn = np.linalg.norm(x)
        ref = np.concatenate([x / nn, [1 / nn]])
        # the float32 fit of absolute coordinates is ill-conditioned far from the origin:
        # tolerance scales with |origin| / spread
        assert np.abs(pabcd[:3] - ref[:3]).max() < tol
        resid = P32.astype(np.float64) @ pabcd[:3].astype(np.float64) + float(pabcd[3])
        assert ok == bool((np.abs(resid) <= 0.4 + 1e-3).all())
        Wd = W.astype(np.float64)
        cs = np.abs(0.5 - Wd).sum()
        assert pcov == pytest.approx((((0.5 - Wd) / cs) ** 2 * Wd).sum(), rel=1e-12)


@pytest.mark.parametrize("origin", [(0, 0, 0), (100, -50, 3), (1000, 800, 10)])
def test_esti_plane_vs_lapack_pivoted_qr_float32(orc, origin):
    """A second, independent implementation of the SAME algorithm class in the SAME precision: LAPACK's sgeqp3
    (Householder QR with column pivoting on the remaining column norms - what Eigen's ColPivHouseholderQR is) solving
    A x = -1 in float32. Agreement here is at float rounding x conditioning, an order of magnitude tighter than the
    double-precision lstsq check above allows far from the origin - it pins the restatement's pivoting / reflector
    arithmetic against an implementation that was not written from the same reading of Eigen."""
    from scipy.linalg import qr, solve_triangular
    rng = np.random.default_rng(4)
    worst = 0.0
    for _ in range(300):
        n = rng.normal(size=3)
        n /= np.linalg.norm(n)
        c = np.asarray(origin, float) + rng.normal(size=3)
        b1 = np.cross(n, rng.normal(size=3))
        b1 /= np.linalg.norm(b1)
        b2 = np.cross(n, b1)
        uv = rng.uniform(-1, 1, (5, 2))
        P32 = (c + uv[:, :1] * b1 + uv[:, 1:] * b2 + rng.normal(0, 0.01, (5, 1)) * n).astype(np.float32)
        ok, pabcd, _ = orc.esti_plane(pts12(P32), 1e9, 0.5)
        Q, R, piv = qr(P32, mode="economic", pivoting=True)          # float32 in, float32 LAPACK
        assert Q.dtype == np.float32
        y = solve_triangular(R, Q.T @ np.full(5, -1, np.float32)).astype(np.float32)
        x = np.zeros(3, np.float32)
        x[piv] = y
        nn = np.float32(np.linalg.norm(x))
        ref = np.concatenate([x / nn, [np.float32(1) / nn]])
        # error scale of a float32 least-squares solve: eps * cond(A); compare relative to it
        cond = np.linalg.cond(P32.astype(np.float64))
        err = np.abs(pabcd[:3] - ref[:3]).max()
        worst = max(worst, err / (6e-8 * cond))
    assert worst < 8.0, worst


def test_esti_plane_flags(orc):
    # a non-planar neighbourhood must be rejected (any residual > plane_th), W[0] <= 1e-5 gives plane_cov 0
    P = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0], [0.5, 0.5, 3.0]], np.float32) + 5
    ok, pabcd, pcov = orc.esti_plane(pts12(P, 0.0), 0.4, 0.5)
    assert not ok and pcov == 0.0
    P[4, 2] = 5.0
    ok, pabcd, _ = orc.esti_plane(pts12(P), 0.4, 0.5)
    assert ok and abs(abs(pabcd[2]) - 1) < 1e-5


# ---- a6: evalPointUncertainty (associate_uct.hpp:153-175) ------------------------------------------
def test_eval_point_uncertainty_vs_numpy(orc, scenes):
    rng = np.random.default_rng(5)
    for _ in range(50):
        A = rng.normal(size=(6, 6))
        cov = 1e-5 * (A @ A.T)
        q = scenes.q_from_rotvec(rng.normal(size=3) * 0.3)
        t = rng.normal(size=3)
        pose = scenes.make_pose(q, t, cov)
        p = rng.uniform(-80, 80, 3).astype(np.float32)
        got = orc.eval_point_uncertainty(pts12(p[None])[0], pose)
        T = pose[7:23].reshape(4, 4)
        p64 = p.astype(np.float64)  # pi.x * distance_weight is a double product (associate_uct.hpp:163)
        pp = T @ np.array([0.05 * p64[0], 0.05 * p64[1], 0.05 * p64[2], 1.0])
        sk = np.array([[0, -pp[2], pp[1]], [pp[2], 0, -pp[0]], [-pp[1], pp[0], 0]])
        G = np.hstack([np.eye(3), -sk, T[:3, :3]])
        S = np.zeros((9, 9))
        S[:6, :6] = 1e4 * cov
        S[6:, 6:] = 0.1 * np.eye(3)
        assert np.allclose(got, G @ S @ G.T, rtol=1e-12, atol=1e-14)


# ---- a15: SE(3) covariance compounding (associate_uct.hpp:29-142) ----------------------------------
def test_compound_pose_and_first_order_cov(orc, scenes):
    rng = np.random.default_rng(7)

    def rnd_pose(scale):
        A = rng.normal(size=(6, 6))
        return scenes.make_pose(scenes.q_from_rotvec(rng.normal(size=3) * 0.5), rng.normal(size=3),
                                scale * (A @ A.T))

    def adj_inv(T):
        R, t = T[:3, :3], T[:3, 3]
        Ri, ti = R.T, -R.T @ t
        sk = np.array([[0, -ti[2], ti[1]], [ti[2], 0, -ti[0]], [-ti[1], ti[0], 0]])
        Ad = np.zeros((6, 6))
        Ad[:3, :3] = Ri
        Ad[:3, 3:] = sk @ Ri
        Ad[3:, 3:] = Ri
        return Ad

    for _ in range(20):
        p1, p2 = rnd_pose(1e-9), rnd_pose(1e-9)
        T1, T2 = p1[7:23].reshape(4, 4), p2[7:23].reshape(4, 4)
        out = orc.compound(p1, p2)
        assert np.allclose(out[7:23].reshape(4, 4), T1 @ T2, atol=1e-12)
        Ad = adj_inv(T2)
        first = Ad @ p1[23:].reshape(6, 6) @ Ad.T + p2[23:].reshape(6, 6)
        assert np.allclose(out[23:].reshape(6, 6), first, rtol=1e-6, atol=1e-18)  # 4th-order terms ~ cov^2
        inv = orc.compound(p1, p2, inverse=True)
        Tc = np.linalg.inv(T1) @ T2
        assert np.allclose(inv[7:23].reshape(4, 4), Tc, atol=1e-12)
        Adc = adj_inv(Tc)
        first = Adc @ p1[23:].reshape(6, 6) @ Adc.T + p2[23:].reshape(6, 6)
        assert np.allclose(inv[23:].reshape(6, 6), first, rtol=1e-6, atol=1e-18)
    # the reference's aliasing quirk (laserMapping.cpp:1043): adjoint taken from the COMPOSED transform
    p1, p2 = rnd_pose(1e-6), rnd_pose(1e-6)
    a, b = orc.compound(p1, p2, alias=True), orc.compound(p1, p2, alias=False)
    assert np.allclose(a[:23], b[:23]) and not np.allclose(a[23:], b[23:], rtol=1e-6)
    # 4th-order terms are really there at larger covariances
    big = orc.compound(rnd_pose(1e-2), rnd_pose(1e-2))
    assert np.isfinite(big).all()


# ---- manifold ops (SOn.hpp, S2.hpp, vect.hpp) --------------------------------------------------------
@pytest.mark.parametrize("L", [1, 2, 3])
def test_boxplus_boxminus_roundtrip(orc, scenes, L):
    rng = np.random.default_rng(11)
    sc_q = [scenes.q_from_rotvec(rng.normal(size=3) * 0.4) for _ in range(L)]
    g = rng.normal(size=3)
    g *= 9.809 / np.linalg.norm(g)
    x = scenes.pack_state(rng.normal(size=3), scenes.q_from_rotvec(rng.normal(size=3)), sc_q, rng.normal(size=(L, 3)),
                          rng.normal(size=3), rng.normal(size=3) * 0.01, rng.normal(size=3) * 0.01, g)
    n = 17 + 6 * L
    d = rng.normal(size=n) * 0.05
    y = orc.boxplus(x, L, d)
    back = orc.boxminus(y, x, L)
    assert np.allclose(back, d, atol=1e-9)
    u = scenes.unpack_state(y, L)
    assert np.linalg.norm(u["grav"]) == pytest.approx(9.809, rel=1e-12)
    # SO3 boxplus == right-multiplication by exp (SOn.hpp:241-244) checked against SciPy
    r0 = Rotation.from_quat(scenes.unpack_state(x, L)["rot"])
    r1 = Rotation.from_quat(u["rot"])
    assert np.allclose((r0 * Rotation.from_rotvec(d[3:6])).as_matrix(), r1.as_matrix(), atol=1e-12)


# ---- a14: SE(3) B-spline (BsplineSE3.cpp) ----------------------------------------------------------------
def test_spline_vs_scipy(orc, scenes):
    rng = np.random.default_rng(13)
    t0 = 1671631987.6  # City dataset epoch: exercises the absolute-double time arithmetic
    ts = t0 + np.arange(0, 0.25, 0.005)
    traj = []
    for k, t in enumerate(ts):
        q = scenes.q_from_rotvec(np.array([0.02, -0.01, 0.3]) * (t - t0) * 4 + rng.normal(size=3) * 1e-3)
        p = np.array([5.0, 0.3, 0.0]) * (t - t0) + rng.normal(size=3) * 1e-3
        traj.append([t, *p, *q])
    traj = np.array(traj)
    sp = orc.Spline(traj)
    ct, cT = sp.control()
    assert len(ct) > 10 and np.allclose(np.diff(ct), 0.01, atol=1e-6)
    # control poses are SE(3) lerps of the samples that bound them (BsplineSE3.cpp:60-77); the last
    # trajectory sample is dropped (:39)
    assert ct[-1] < traj[-2, 0] + 1e-9
    for tq in t0 + np.array([0.0311, 0.0777, 0.1234, 0.15]):
        ok, q, p = sp.get_pose(tq)
        assert ok
        i1 = np.searchsorted(ct, tq, side="right") - 1
        T = [cT[i1 - 1], cT[i1], cT[i1 + 1], cT[i1 + 2]]
        u = (tq - ct[i1]) / (ct[i1 + 1] - ct[i1])
        b = [(5 + 3 * u - 3 * u * u + u ** 3) / 6, (1 + 3 * u + 3 * u * u - 2 * u ** 3) / 6, u ** 3 / 6]
        X = T[0].copy()
        for k in range(3):
            X = X @ expm(b[k] * np.real(logm(np.linalg.inv(T[k]) @ T[k + 1])))
        assert np.allclose(p, X[:3, 3], atol=1e-9)
        assert np.allclose(Rotation.from_quat(q).as_matrix(), X[:3, :3], atol=1e-9)
    assert not sp.get_pose(t0 - 1.0)[0] and not sp.get_pose(ts[-1] + 1.0)[0]
    assert not sp.get_pose(ct[0] + 0.001)[0]  # needs one older control point (BsplineSE3.cpp:199-201)


# ---- a2: exact 5-NN ---------------------------------------------------------------------------------------
def _f32_d2(q, p):
    d = q.astype(np.float32)[:, None, :] - p.astype(np.float32)
    return (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]


def test_knn_kd_vs_ckdtree_and_ref(orc, scenes):
    sc = scenes.make_scene(seed=21, N=1500, Nmap=30000, L=1)
    q = sc["map"][::20].copy()
    q[:, :3] += np.float32(0.13)
    own = orc.Oracle(sc["params"], threads=2, use_ref=False)
    own.map_build(sc["map"])
    assert not own.is_ref
    po, d2o, co = own.knn(q)
    tree = cKDTree(sc["map"][:, :3].astype(np.float64))
    dd, ii = tree.query(q[:, :3].astype(np.float64), k=5)
    ref_d2 = _f32_d2(q[:, :3], sc["map"][ii, :3])
    assert (co == 5).all()
    assert np.array_equal(np.sort(d2o, 1), d2o)  # ascending (ikd_Tree.cpp:452-458)
    # float32 distances of the double-precision neighbour set agree except at float-level ties
    assert np.mean(d2o == np.sort(ref_d2, 1)) > 0.999
    assert np.allclose(d2o, np.sort(ref_d2, 1), rtol=1e-5)
    if orc.have_ref():
        ref = orc.Oracle(sc["params"], threads=2, use_ref=True)
        assert ref.is_ref
        ref.map_build(sc["map"])
        pr, d2r, cr = ref.knn(q)
        assert np.array_equal(d2r, d2o) and np.array_equal(cr, co)  # bit-exact squared distances
        assert np.mean((pr[:, :, :3] == po[:, :, :3]).all(-1)) > 0.999  # same points except exact ties


# ---- a1/a5: Jacobian rows are the derivatives of the residual ---------------------------------------------
@pytest.mark.parametrize("L", [1, 3])
def test_rows_are_residual_jacobian(orc, scenes, L):
    sc = scenes.make_scene(seed=31 + L, N=400, Nmap=20000, L=L)
    o = orc.Oracle(sc["params"], threads=1)
    o.map_build(sc["map"])
    o.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"])
    x0 = sc["state0"]
    r0 = o.h_share_model(x0, True)
    sel = o.scan_get()["selected"].astype(bool)
    assert r0["M"] == sel.sum() > 300
    # undo the scalings (c_i and w) to get the raw row: h = -pd2 * c * w  ->  scale_i = h_i / (-pd2_i)
    pd2 = o.scan_get()["normvec"][sel, 3].astype(np.float64)
    scale = r0["h"] / (-pd2)
    n = 17 + 6 * L
    C = 6 * (1 + L)
    eps = 2e-3
    for col in range(C):
        d = np.zeros(n)
        # tangent index of H column `col`: [pos, rot, R_0..R_{L-1}, T_0..T_{L-1}] are the first C tangent dims
        d[col] = eps
        xp, xm = orc.boxplus(x0, L, d), orc.boxplus(x0, L, -d)
        o.h_share_model(xp, False)
        gp = o.scan_get()
        o.h_share_model(xm, False)
        gm = o.scan_get()
        both = gp["selected"].astype(bool) & gm["selected"].astype(bool) & sel
        num = (gp["normvec"][:, 3].astype(np.float64) - gm["normvec"][:, 3]) / (2 * eps)
        ana = np.zeros(len(sel))
        ana[sel] = r0["h_x"][:, col] / scale
        assert both.sum() > 250
        err = np.abs(num[both] - ana[both])
        assert err.max() < 2e-2 * max(1.0, np.abs(ana[both]).max()), (col, err.max())
        o.h_share_model(x0, False)  # restore flags for the next column (reuse passes only ever drop points)
        o.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"])
        o.h_share_model(x0, True)


# ---- a10/a11: the iterated update converges to the ground truth of the synthetic scene ------------------
@pytest.mark.parametrize("L,kind", [(1, "city"), (2, "city"), (3, "city"), (3, "tunnel")])
def test_update_converges(orc, scenes, L, kind):
    sc = scenes.make_scene(seed=41 + L, N=3000, Nmap=60000, L=L, kind=kind, det_range=100.0 if kind == "city" else 500)
    o = orc.Oracle(sc["params"], threads=2)
    o.map_build(sc["map"])
    o.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"])
    u = o.update_iterated(sc["state0"], sc["P0"])
    assert 2 <= u["passes"] <= sc["params"]["max_iteration"] + 1 and u["M"] > 2000
    g, s0, s1 = (scenes.unpack_state(v, L) for v in (sc["state_gt"], sc["state0"], u["state"]))
    if kind == "city":
        assert np.linalg.norm(s1["pos"] - g["pos"]) < 0.25 * np.linalg.norm(s0["pos"] - g["pos"])
    else:  # the tunnel axis is unobservable: only the cross-axis error must shrink
        assert np.linalg.norm((s1["pos"] - g["pos"])[1:]) < 0.25 * np.linalg.norm((s0["pos"] - g["pos"])[1:])
    P = u["P"]
    # P = L - K_x P with differently projected L and P (esekfom.hpp:667-714) is only approximately symmetric
    dg = np.sqrt(np.abs(np.diag(P)))
    assert (np.abs(P - P.T) <= 0.05 * np.outer(dg, dg) + 1e-15).all()
    assert np.linalg.eigvalsh(0.5 * (P + P.T)).min() > -1e-9
    assert P[0, 0] < sc["P0"][0, 0]


def test_localization_weight_branches(orc, scenes):
    # plain (ground only): sigma3/sigma1 below localize_thresh_min -> localize_cov_min; the 10 m x 6 m tunnel
    # lands in between because plane fits across its corners tilt along the axis; city: rich geometry
    for kind, dr in (("plain", 100.0), ("tunnel", 500.0), ("city", 100.0)):
        sc = scenes.make_scene(seed=51, N=3000, Nmap=60000, L=1, kind=kind, det_range=dr)
        o = orc.Oracle(sc["params"], threads=2)
        o.map_build(sc["map"])
        o.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"])
        r = o.h_share_model(sc["state_gt"], True)
        raw = r["h_x"][:, :3] / r["weight"]
        sv = np.linalg.svd(raw, compute_uv=False)
        ratio = sv[2] / sv[0]
        p = sc["params"]
        if ratio > p["localize_thresh_max"]:
            want = p["localize_cov_max"]
        elif ratio < p["localize_thresh_min"]:
            want = p["localize_cov_min"]
        else:
            want = ((p["localize_cov_max"] - p["localize_cov_min"]) * (ratio - p["localize_thresh_min"]) /
                    (p["localize_thresh_max"] - p["localize_thresh_min"]) + p["localize_cov_min"])
        assert r["weight"] == pytest.approx(want, rel=1e-9)
        if kind == "plain":
            assert ratio < p["localize_thresh_min"] and r["weight"] == p["localize_cov_min"]


def test_update_limit_and_pass_hook(orc, scenes):
    """esekf's `limit` (esekfom.hpp:160-163): tightened so that nothing converges the loop runs max_iteration + 1 passes
    with the search forced at i == maximum_iter - 2 (:660-663); a pass hook that empties the map after pass 0 leaves
    the state of pass 0 and the PROJECTED prior covariance (:514-531)."""
    kw = dict(seed=231, N=600, Nmap=20000, L=3, kind="tunnel", det_range=500.0, max_iteration=9)
    sc = scenes.make_scene(limit=1e-30, **kw)
    o = orc.Oracle(sc["params"], threads=2)
    o.map_build(sc["map"]), o.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"])
    v = o.update_iterated(sc["state0"], sc["P0"])
    assert (v["passes"], v["searches"]) == (10, 2)
    sc3 = scenes.make_scene(**kw)
    o3 = orc.Oracle(sc3["params"], threads=2)
    o3.map_build(sc3["map"]), o3.scan_set(sc3["scan"], sc3["tables"], sc3["temporal_comp"])
    assert o3.update_iterated(sc3["state0"], sc3["P0"])["passes"] < 10

    sc = scenes.make_scene(seed=232, N=600, Nmap=20000, L=2, prior_dpos=0.0, prior_drot_deg=0.0, limit=0.05)  # pass 0 converges
    o = orc.Oracle(sc["params"], threads=2)
    o.map_build(sc["map"]), o.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"])
    far = sc["map"][:64].copy()
    far[:, 0] += 5000.0
    seen = []
    o.set_pass_hook(lambda k: (seen.append(k), o.map_build(far) if k == 1 else None))
    w = o.update_iterated(sc["state0"], sc["P0"])
    o.set_pass_hook(None)
    assert seen == [0, 1, 2, 3] and w["passes"] == 4 and w["searches"] == 4
    assert not np.array_equal(w["state"], sc["state0"]) and not np.array_equal(w["P"], sc["P0"])
    assert np.abs(w["P"] - sc["P0"]).max() < 1e-3 * np.abs(sc["P0"]).max()  # the projection is close to the identity
