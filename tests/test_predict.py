"""Row f-3: one esekf::predict step (esekfom.hpp:388-492 + use-ikfom.hpp:67-112).

Three layers, all CPU (malio_predict is pure host code):
  * the oracle's dense restatement against an independent NumPy/SciPy construction of F, G and the flow, and that
    construction against finite differences of the discrete map (the reference's F is first order in dt and keeps the
    identity where exp(-w dt) would stand - the integer 1/2 of esekfom.hpp:421 - so that check is O(dt));
  * the product's banded O(n^2) form against the oracle;
  * long chains, state-only calls, argument errors.
Parity with the reference binary is unpinned (the IKFoM headers need Eigen and boost, neither is in this image)."""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation as Rot

G_LEN = 9.809


def rnd_state(scenes, rng, L, small_bias=True):
    g = rng.normal(size=3)
    g = g / np.linalg.norm(g) * G_LEN
    if g[0] < -9.0:
        g[0] = -g[0]
    offR = [scenes.q_from_rotvec(rng.normal(size=3) * 0.5) for _ in range(L)]
    offT = [rng.normal(size=3) for _ in range(L)]
    b = 0.01 if small_bias else 1.0
    return scenes.pack_state(rng.normal(size=3) * 10, scenes.q_from_rotvec(rng.normal(size=3)), offR, offT,
                             vel=rng.normal(size=3) * 3, bg=rng.normal(size=3) * b, ba=rng.normal(size=3) * b, grav=g)


def rnd_cov(rng, n, scale):
    A = rng.normal(size=(n, n))
    return scale * (A @ A.T) / n


def rnd_Q(rng):
    Q = np.zeros((12, 12))
    Q[0:3, 0:3] = np.eye(3) * 1e-2
    Q[3:6, 3:6] = np.eye(3) * 1e-1
    Q[6:9, 6:9] = np.eye(3) * 1e-4
    Q[9:12, 9:12] = np.eye(3) * 1e-3
    B = rng.normal(size=(12, 12)) * 1e-3
    return Q + B @ B.T


def hat(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])


def unpack(x, L):
    o = 0
    d = {}
    d["pos"], o = x[o:o + 3], o + 3
    d["rot"], o = x[o:o + 4], o + 4
    d["offR"], o = x[o:o + 4 * L].reshape(L, 4), o + 4 * L
    d["offT"], o = x[o:o + 3 * L].reshape(L, 3), o + 3 * L
    d["vel"], o = x[o:o + 3], o + 3
    d["bg"], o = x[o:o + 3], o + 3
    d["ba"], o = x[o:o + 3], o + 3
    d["grav"] = x[o:o + 3]
    return d


def s2_Bx(g):
    d = G_LEN + g[0]
    B = np.array([[-g[1], -g[2]], [G_LEN - g[1] * g[1] / d, -g[2] * g[1] / d], [-g[2] * g[1] / d, G_LEN - g[2] * g[2] / d]])
    return B / G_LEN


def A_matrix(v):
    n = np.linalg.norm(v)
    if n < 1e-11:
        return np.eye(3)
    H = hat(v)
    return np.eye(3) + (1 - np.cos(n)) / n**2 * H + (1 - np.sin(n) / n) / n**2 * (H @ H)


def numpy_predict(x, P, dt, Q, acc, gyro, L):
    """Independent construction: whole F and G assembled from the block formulas, dense products."""
    n = 17 + 6 * L
    s = unpack(np.array(x, np.float64), L)
    iv = 6 * (L + 1)
    R = Rot.from_quat(s["rot"]).as_matrix()
    w = gyro - s["bg"]
    a = acc - s["ba"]
    Bx = s2_Bx(s["grav"])
    F = np.eye(n)
    F[0:3, iv:iv + 3] += dt * np.eye(3)
    F[3:6, iv + 3:iv + 6] += dt * (-A_matrix(-w * dt))
    F[iv:iv + 3, 3:6] += dt * (-R @ hat(a))
    F[iv:iv + 3, iv + 6:iv + 9] += dt * (-R)
    F[iv:iv + 3, iv + 9:iv + 11] += dt * (-hat(s["grav"]) @ Bx)
    Nx = Bx.T @ hat(s["grav"]) / G_LEN**2
    F[iv + 9:iv + 11, iv + 9:iv + 11] = Nx @ (-hat(s["grav"]) @ Bx)
    G = np.zeros((n, 12))
    G[3:6, 0:3] = dt * (-A_matrix(-w * dt))
    G[iv:iv + 3, 3:6] = dt * (-R)
    G[iv + 3:iv + 6, 6:9] = dt * np.eye(3)
    G[iv + 6:iv + 9, 9:12] = dt * np.eye(3)
    xo = np.array(x, np.float64)
    so = unpack(xo, L)
    so["pos"] += dt * s["vel"]
    so["rot"][:] = (Rot.from_quat(s["rot"]) * Rot.from_rotvec(w * dt)).as_quat()
    so["vel"] += dt * (R @ a + s["grav"])
    return xo, F @ P @ F.T + G @ Q @ G.T, F


def quat_close(a, b, tol):
    return min(np.abs(a - b).max(), np.abs(a + b).max()) < tol


@pytest.mark.parametrize("L", [1, 2, 3, 4])
def test_oracle_matches_numpy(orc, scenes, L):
    rng = np.random.default_rng(100 + L)
    n = 17 + 6 * L
    for k in range(10):
        x = rnd_state(scenes, rng, L)
        P = rnd_cov(rng, n, 1e-3)
        dt = [1e-3, 5e-3, 1e-2, 0.1][k % 4]
        acc, gyro = rng.normal(size=3) * 5 + [0, 0, 9.8], rng.normal(size=3)
        Q = rnd_Q(rng)
        xo, Po = orc.predict(L, x, P, dt, Q, acc, gyro)
        xn, Pn, _ = numpy_predict(x, P, dt, Q, acc, gyro, L)
        so, sn = unpack(xo, L), unpack(xn, L)
        for key in ("pos", "offT", "vel", "bg", "ba", "grav"):
            assert np.allclose(so[key], sn[key], rtol=0, atol=1e-12), key
        assert quat_close(so["rot"], sn["rot"], 1e-13)
        assert np.array_equal(so["offR"], unpack(x, L)["offR"])
        assert np.allclose(Po, Pn, rtol=1e-11, atol=1e-16)


def boxplus(scenes, x, d, L):
    s = unpack(np.array(x, np.float64), L)
    out = np.array(x, np.float64)
    o = unpack(out, L)
    iv = 6 * (L + 1)
    o["pos"] += d[0:3]
    o["rot"][:] = (Rot.from_quat(s["rot"]) * Rot.from_rotvec(d[3:6])).as_quat()
    for l in range(L):
        o["offR"][l] = (Rot.from_quat(s["offR"][l]) * Rot.from_rotvec(d[6 + 3 * l:9 + 3 * l])).as_quat()
        o["offT"][l] += d[6 + 3 * L + 3 * l:9 + 3 * L + 3 * l]
    o["vel"] += d[iv:iv + 3]
    o["bg"] += d[iv + 3:iv + 6]
    o["ba"] += d[iv + 6:iv + 9]
    Bu = s2_Bx(s["grav"]) @ d[iv + 9:iv + 11]
    o["grav"][:] = Rot.from_rotvec(Bu).as_matrix() @ s["grav"]
    return out


def boxminus(a, b, L):
    sa, sb = unpack(np.array(a, np.float64), L), unpack(np.array(b, np.float64), L)
    n = 17 + 6 * L
    iv = 6 * (L + 1)
    d = np.zeros(n)
    d[0:3] = sa["pos"] - sb["pos"]
    d[3:6] = (Rot.from_quat(sb["rot"]).inv() * Rot.from_quat(sa["rot"])).as_rotvec()
    for l in range(L):
        d[6 + 3 * l:9 + 3 * l] = (Rot.from_quat(sb["offR"][l]).inv() * Rot.from_quat(sa["offR"][l])).as_rotvec()
        d[6 + 3 * L + 3 * l:9 + 3 * L + 3 * l] = sa["offT"][l] - sb["offT"][l]
    d[iv:iv + 3] = sa["vel"] - sb["vel"]
    d[iv + 3:iv + 6] = sa["bg"] - sb["bg"]
    d[iv + 6:iv + 9] = sa["ba"] - sb["ba"]
    g, o = sa["grav"], sb["grav"]
    c = np.cross(g, o)
    vs = np.linalg.norm(c)
    if vs > 1e-11:
        d[iv + 9:iv + 11] = np.arctan2(vs, g @ o) / vs * (s2_Bx(o).T @ np.cross(o, g))
    return d


def test_jacobian_is_first_order_in_dt(orc, scenes):
    """F of the reference against central differences of delta -> predict(x [+] delta) [-] predict(x)."""
    L = 2
    n = 17 + 6 * L
    rng = np.random.default_rng(7)
    x = rnd_state(scenes, rng, L)
    acc, gyro = np.array([0.3, -0.2, 9.7]), np.array([0.2, -0.1, 0.3])
    Q, P = np.zeros((12, 12)), np.eye(n)
    for dt in (1e-2, 1e-3):
        _, _, F = numpy_predict(x, P, dt, Q, acc, gyro, L)
        x1, _ = orc.predict(L, x, P, dt, Q, acc, gyro)
        eps = 1e-6
        J = np.zeros((n, n))
        for j in range(n):
            e = np.zeros(n)
            e[j] = eps
            xp, _ = orc.predict(L, boxplus(scenes, x, e, L), P, dt, Q, acc, gyro)
            xm, _ = orc.predict(L, boxplus(scenes, x, -e, L), P, dt, Q, acc, gyro)
            J[:, j] = (boxminus(xp, x1, L) - boxminus(xm, x1, L)) / (2 * eps)
        # exact Jacobian has exp(-w dt) where F keeps I, and second-order terms elsewhere: both O(dt)
        assert np.abs(J - F).max() < 1.5 * dt, (dt, np.abs(J - F).max())
        assert np.abs(J - F).max() > 1e-3 * dt  # ... and the difference is real: F is not the exact Jacobian


@pytest.mark.parametrize("L", [1, 2, 3, 4])
def test_product_matches_oracle(capi, orc, scenes, L):
    rng = np.random.default_rng(200 + L)
    n = 17 + 6 * L
    for k in range(25):
        x = rnd_state(scenes, rng, L, small_bias=(k % 2 == 0))
        P = rnd_cov(rng, n, [1e-6, 1e-3, 1.0][k % 3])
        dt = [2.5e-3, 5e-3, 1e-2, 0.05, 0.0][k % 5]
        acc, gyro = rng.normal(size=3) * 5 + [0, 0, 9.8], rng.normal(size=3) * [1, 1, 1e-3][k % 3]
        Q = rnd_Q(rng)
        xg, Pg = capi.predict(L, x, P, dt, Q, acc, gyro)
        xo, Po = orc.predict(L, x, P, dt, Q, acc, gyro)
        assert np.allclose(xg, xo, rtol=0, atol=1e-13 * max(1.0, np.abs(xo).max()))
        assert np.allclose(Pg, Po, rtol=1e-12, atol=1e-18 + 1e-14 * np.abs(Po).max())
        assert np.allclose(Pg, Pg.T, rtol=1e-12, atol=1e-18 + 1e-14 * np.abs(Po).max())


def test_zero_rate_and_gravity_on_axis(capi, orc, scenes):
    """gyro == bg (A_matrix's small-angle branch) and gravity along -z / +x (S2_Bx's regular branch at its edges)."""
    L = 1
    n = 17 + 6 * L
    rng = np.random.default_rng(5)
    for grav in ([0, 0, -G_LEN], [G_LEN, 0, 0], [0, G_LEN, 0]):
        x = scenes.pack_state([1, 2, 3], scenes.q_from_rotvec(np.array([0.1, 0.2, 0.3])), [scenes.q_from_rotvec(np.zeros(3))],
                              [np.zeros(3)], vel=[1, 0, 0], bg=[0.01, 0.02, 0.03], ba=[0, 0, 0], grav=grav)
        P = rnd_cov(rng, n, 1e-2)
        xg, Pg = capi.predict(L, x, P, 0.01, rnd_Q(rng), [0, 0, 9.8], [0.01, 0.02, 0.03])
        xo, Po = orc.predict(L, x, P, 0.01, rnd_Q(np.random.default_rng(5)), [0, 0, 9.8], [0.01, 0.02, 0.03])
        assert np.allclose(xg, xo, rtol=0, atol=1e-13)
        assert np.array_equal(unpack(xg, L)["rot"], unpack(np.array(x, np.float64), L)["rot"])
    # same Q for the covariance comparison
    Q = rnd_Q(rng)
    xg, Pg = capi.predict(L, x, P, 0.01, Q, [0, 0, 9.8], [0.01, 0.02, 0.03])
    xo, Po = orc.predict(L, x, P, 0.01, Q, [0, 0, 9.8], [0.01, 0.02, 0.03])
    assert np.allclose(Pg, Po, rtol=1e-12, atol=1e-18)


def test_chain_like_forward_propagation(capi, orc, scenes):
    """IMU_Processing.hpp:305-345: 400 steps at 200 Hz between two scans' worth of IMU, Q constant."""
    L = 2
    n = 17 + 6 * L
    rng = np.random.default_rng(11)
    xg = xo = rnd_state(scenes, rng, L)
    Pg = Po = np.eye(n) * 1e-4
    Q = rnd_Q(rng)
    for k in range(400):
        t = k * 0.005
        acc = np.array([np.sin(t) * 2, np.cos(2 * t), 9.8 + 0.3 * np.sin(3 * t)])
        gyro = np.array([0.3 * np.cos(t), 0.2 * np.sin(2 * t), 0.5])
        xg, Pg = capi.predict(L, xg, Pg, 0.005, Q, acc, gyro)
        xo, Po = orc.predict(L, xo, Po, 0.005, Q, acc, gyro)
    assert np.allclose(xg, xo, rtol=0, atol=1e-10)
    assert np.allclose(Pg, Po, rtol=1e-9, atol=1e-15)
    assert np.linalg.eigvalsh((Pg + Pg.T) / 2).min() > 0
    assert abs(np.linalg.norm(unpack(xg, L)["rot"]) - 1) < 1e-12


def test_state_only_and_errors(capi, orc, scenes):
    import ctypes as C
    L = 3
    rng = np.random.default_rng(3)
    x = rnd_state(scenes, rng, L)
    acc, gyro, Q = np.array([0.1, 0.2, 9.7]), np.array([0.1, 0.0, -0.2]), rnd_Q(rng)
    x1, none = capi.predict(L, x, None, 0.01, Q, acc, gyro)
    x2, _ = capi.predict(L, x, np.eye(17 + 6 * L), 0.01, Q, acc, gyro)
    assert none is None and np.array_equal(x1, x2)
    lib = capi.lib()
    s = capi.state_from_flat(x, L)
    a = (C.c_double * 3)(*acc)
    P = (C.c_double * (35 * 35))()
    assert lib.malio_predict(0, C.byref(s), None, C.c_double(0.01), None, a, a) == capi.ERR_BAD_ARG
    assert lib.malio_predict(9, C.byref(s), None, C.c_double(0.01), None, a, a) == capi.ERR_BAD_ARG
    assert lib.malio_predict(L, None, None, C.c_double(0.01), None, a, a) == capi.ERR_BAD_ARG
    assert lib.malio_predict(L, C.byref(s), None, C.c_double(0.01), None, None, a) == capi.ERR_BAD_ARG
    assert lib.malio_predict(L, C.byref(s), P, C.c_double(0.01), None, a, a) == capi.ERR_BAD_ARG  # P without Q


def test_golden_chain(capi, orc):
    """tests/golden/predict_chain.npz (tests/golden/make_golden.py predict): the committed pin of the propagation step -
    the oracle must keep reproducing it bit for bit, the product's banded form to rounding."""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "predict_chain.npz"))
    L, dt, Q = int(z["L"]), float(z["dt"]), z["Q"]
    xo = xg = z["x_start"]
    Po = Pg = z["P0"]
    for k in range(z["acc"].shape[0]):
        xo, Po = orc.predict(L, xo, Po, dt, Q, z["acc"][k], z["gyro"][k])
        xg, Pg = capi.predict(L, xg, Pg, dt, Q, z["acc"][k], z["gyro"][k])
        assert np.array_equal(xo, z["x"][k])
        assert np.allclose(xg, z["x"][k], rtol=0, atol=1e-11)
        if k == 9:
            assert np.array_equal(Po, z["P_10"])
            assert np.allclose(Pg, z["P_10"], rtol=1e-10, atol=1e-18)
    assert np.array_equal(Po, z["P_50"])
    assert np.allclose(Pg, z["P_50"], rtol=1e-10, atol=1e-18)


@pytest.mark.gpu
@pytest.mark.parametrize("L", [1, 3, 4])
def test_device_chain_equals_host_chain(capi, orc, scenes, L):
    """malio_predict_chain (row f-3 on the device): three tracks of different lengths side by side - what
    ImuProcess::UndistortPcl runs one after the other as kf.predict / predict_cont / back_predict - against the ORACLE's
    dense restatement of esekf::predict (esekfom.hpp:388-492 with MA-LIO's process model, use-ikfom.hpp:67-112) stepped
    alongside: the state after EVERY step (1e-12) and the covariance at the end of every track (1e-10 relative) - and,
    second, against the same steps taken one by one with the host's malio_predict (same operation order; the device's
    sin / cos may differ from glibc's in the last place)."""
    rng = np.random.default_rng(40 + L)
    n = 17 + 6 * L
    sc = scenes.make_scene(seed=12, N=200, Nmap=3000, L=L)
    eng = capi.Engine(sc["params"], device=0)
    Ks = [40, 17, 1]
    xs = [rnd_state(scenes, rng, L, small_bias=(t == 1)) for t in range(3)]
    Ps = [rnd_cov(rng, n, [1e-4, 1e-2, 1.0][t]) for t in range(3)]
    Q = rnd_Q(rng)
    dts, accs, gyros = [], [], []
    for t, K in enumerate(Ks):
        tt = np.arange(K) * 0.005
        dts.append(np.full(K, [0.005, 0.01, 0.0025][t]))
        accs.append(np.stack([np.sin(tt) * 2, np.cos(2 * tt), 9.8 + 0.3 * np.sin(3 * tt)], 1))
        gyros.append(np.stack([0.3 * np.cos(tt), 0.2 * np.sin(2 * tt), np.full(K, 0.5 if t != 2 else 0.0)], 1))
    ends, Pe, steps = eng.predict_chain(xs, Ps, dts, accs, gyros, Q)
    for t, K in enumerate(Ks):
        x, P = xs[t], Ps[t]
        xo, Po = xs[t], Ps[t]
        for k in range(K):
            xo, Po = orc.predict(L, xo, Po, dts[t][k], Q, accs[t][k], gyros[t][k])  # the oracle: parity proper
            assert np.allclose(steps[t][k], xo, rtol=0, atol=1e-12 * max(1.0, np.abs(xo).max())), ("oracle", t, k)
            x, P = capi.predict(L, x, P, dts[t][k], Q, accs[t][k], gyros[t][k])
            assert np.allclose(steps[t][k], x, rtol=0, atol=1e-12 * max(1.0, np.abs(x).max())), (t, k)
        assert np.allclose(ends[t], xo, rtol=0, atol=1e-12 * max(1.0, np.abs(xo).max()))
        assert np.allclose(Pe[t], Po, rtol=1e-10, atol=1e-18 + 1e-13 * np.abs(Po).max()), ("oracle P", t)
        assert np.allclose(ends[t], x, rtol=0, atol=1e-12 * max(1.0, np.abs(x).max()))
        assert np.allclose(Pe[t], P, rtol=1e-10, atol=1e-18 + 1e-13 * np.abs(P).max())
    # states only (P == NULL), one track
    e2, none, s2 = eng.predict_chain(xs[:1], None, dts[:1], accs[:1], gyros[:1], Q)
    assert none is None and np.allclose(e2[0], ends[0], rtol=0, atol=1e-12 * max(1.0, np.abs(ends[0]).max()))
