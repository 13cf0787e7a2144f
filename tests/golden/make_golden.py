"""Generates tests/golden/*.npz: small seeded problems with the outputs of the CPU oracle, whose k-NN is
the REFERENCE's own ikd-Tree (oracle/_ref, compiled from /root/reference) when available.
The reference ships no golden vectors, so these are the committed pins:
    python tests/golden/make_golden.py          (run in the build container, /root/reference present)
Inputs are stored too, so neither the GPU box nor CI needs the generator or /root/reference."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

ge.load_package()
from malio_amd import scenes  # noqa: E402
from oracle import orc  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = {
    "city_L3": dict(seed=101, N=600, Nmap=12000, L=3, kind="city"),
    "urban_L2_unc": dict(seed=102, N=500, Nmap=10000, L=2, kind="city", map_unc=True),
    "velodyne_L1_noext": dict(seed=103, N=400, Nmap=8000, L=1, kind="city", extrinsic_est_en=0),
    "tunnel_L3": dict(seed=104, N=500, Nmap=12000, L=3, kind="tunnel", det_range=500.0, max_iteration=9),
}


def main():
    assert orc.have_ref(), "build oracle/_ref first (make -C oracle)"
    for name, kw in CASES.items():
        sc = scenes.make_scene(**kw)
        o = orc.Oracle(sc["params"], threads=1, use_ref=True)
        assert o.is_ref
        o.map_build(sc["map"])
        o.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"])
        r = o.h_share_model(sc["state0"], True)
        g = o.scan_get()
        s2 = sc["state0"].copy()
        s2[0:3] += [0.01, -0.02, 0.005]
        r2 = o.h_share_model(s2, False)
        g2 = o.scan_get()
        o.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"])
        u = o.update_iterated(sc["state0"], sc["P0"])
        q = sc["map"][:: max(1, sc["Nmap"] // 300)].copy()
        q[:, :3] += np.float32(0.07)
        kp, kd2, kc = o.knn(q)
        out = dict(
            params=np.array([sc["params"][k] for k in orc.PARAM_ORDER], np.float64), map=sc["map"], scan=sc["scan"],
            tables=np.concatenate(sc["tables"], 0), table_len=np.array([t.shape[0] for t in sc["tables"]], np.int32),
            temporal_comp=sc["temporal_comp"], state0=sc["state0"], P0=sc["P0"],
            p1_M=r["M"], p1_hx=r["h_x"], p1_h=r["h"], p1_R=r["R"], p1_weight=r["weight"], p1_selected=g["selected"],
            p1_normvec=g["normvec"], p1_normal_y=g["normal_y"], p1_world=g["world"], p1_nearest=g["nearest"][:, :, :3],
            p1_nearest_cnt=g["nearest_cnt"], state2=s2, p2_M=r2["M"], p2_hx=r2["h_x"], p2_h=r2["h"], p2_R=r2["R"],
            p2_selected=g2["selected"], upd_state=u["state"], upd_P=u["P"], upd_passes=u["passes"],
            upd_searches=u["searches"], upd_M=u["M"], knn_q=q, knn_d2=kd2, knn_cnt=kc, knn_xyz=kp[:, :, :3])
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, "M", r["M"], "passes", u["passes"], os.path.getsize(os.path.join(HERE, name + ".npz")) // 1024, "KiB")


def predict_fixture():
    """tests/golden/predict_chain.npz: 50 steps of esekf::predict (oracle restatement) on a 3-LiDAR state, 200 Hz."""
    L = 3
    sc = scenes.make_scene(seed=105, N=200, Nmap=3000, L=L)
    rng = np.random.default_rng(105)
    x = np.array(sc["state0"], np.float64)
    iv = 7 + 7 * L                                       # flat index of vel (quaternions take 4)
    x[iv:iv + 3] = [1.5, -0.4, 0.1]
    x[iv + 3:iv + 9] = rng.normal(size=6) * 0.01         # bg, ba
    x_start = x.copy()
    P = np.array(sc["P0"], np.float64)
    Q = np.diag([1e-2] * 3 + [1e-1] * 3 + [1e-4] * 3 + [1e-3] * 3)
    accs, gyros, xs, P10 = [], [], [], None
    for k in range(50):
        t = k * 0.005
        acc = np.array([np.sin(t) * 2, np.cos(2 * t), 9.8 + 0.3 * np.sin(3 * t)])
        gyro = np.array([0.3 * np.cos(t), 0.2 * np.sin(2 * t), 0.5])
        x, P = orc.predict(L, x, P, 0.005, Q, acc, gyro)
        accs.append(acc), gyros.append(gyro), xs.append(x.copy())
        if k == 9:
            P10 = P.copy()
    np.savez_compressed(os.path.join(HERE, "predict_chain.npz"), L=L, x_start=x_start, P0=np.array(sc["P0"], np.float64),
                        Q=Q, dt=0.005, acc=np.array(accs), gyro=np.array(gyros), x=np.array(xs), P_10=P10, P_50=P)
    print("predict_chain", os.path.getsize(os.path.join(HERE, "predict_chain.npz")) // 1024, "KiB")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "predict":
        predict_fixture()     # python tests/golden/make_golden.py predict   (leaves the scene fixtures untouched)
    else:
        main()
