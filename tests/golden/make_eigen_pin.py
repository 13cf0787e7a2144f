"""Inputs for, and golden outputs from, the reference's own Eigen-typed code (oracle/ref_eigen).

build_inputs() is deterministic (seeded scenes + the oracle only to RECORD measurement rows as replay inputs; the
quantities under test are recomputed by both sides from these inputs). Run as a script - `make -C oracle/ref_eigen golden`
- on a machine that has Eigen 3 + Boost and /root/reference: runs oracle/_ref/ref_eigen on the inputs and writes
tests/golden/eigen_pin.npz (inputs + reference outputs). tests/test_eigen_pin.py then compares the oracle against it
wherever the suite runs; without the file the test skips and parity stays "unpinned"."""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import eigen_io  # noqa: E402

GOLDEN = os.path.join(HERE, "eigen_pin.npz")
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "ref_eigen")


def build_inputs():
    import __graft_entry__ as ge
    ge.load_package()
    from malio_amd import scenes
    from oracle import orc
    rng = np.random.default_rng(20230625)
    sc = scenes.make_scene(seed=7001, N=600, Nmap=20000, L=3)  # the reference binary is compiled for lid_num = 3
    o = orc.Oracle(sc["params"], threads=1)
    o.map_build(sc["map"])
    o.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"])
    inp = {}
    # a3: neighbour sets as the search returns them (real planes), far from / near the origin, and degenerate ones
    r0 = o.h_share_model(sc["state0"], True)
    near = o.scan_get()["nearest"]  # [N,5,12]
    pts = near[:, :, [0, 1, 2, 5]].astype(np.float32)
    deg = pts[:40].copy()
    deg[:, :, 2] = deg[:, :1, 2]                 # exactly coplanar in z
    deg[:20, 1:, :3] = deg[:20, :1, :3]          # coincident points: rank-deficient A
    unc = pts[:100].copy()
    unc[:, :, 3] = rng.uniform(0, 0.002, (100, 5)).astype(np.float32)  # map uncertainty on (common_lib.h:159-173)
    inp["plane_pts"] = np.concatenate([pts, pts[:200] + np.float32(1000.0) * np.array([1, -1, 0, 0], np.float32), deg, unc])
    inp["plane_th"] = np.array([sc["params"]["plane_th"]], np.float32)
    inp["cov_threshold"] = np.array([sc["params"]["cov_threshold"]], np.float64)
    # a6: points x table entries
    tab = np.concatenate([np.asarray(t, np.float64).reshape(-1, 59) for t in sc["tables"]])
    k = rng.integers(0, tab.shape[0], 300)
    inp["unc_pts"] = sc["scan"][:300, 0:3].astype(np.float32)
    inp["unc_poses"] = tab[k]
    # a15: compounding of table entries (covariances as the scene draws them)
    ka, kb = rng.integers(0, tab.shape[0], 64), rng.integers(0, tab.shape[0], 64)
    inp["comp_a"], inp["comp_b"] = tab[ka], tab[kb]
    # a10-a12: recorded measurement rows for max_iteration + 1 passes (pass 2 invalid, pass 3 with M < n)
    max_iter = 4
    passes = []
    for p in range(max_iter + 1):
        s = sc["state0"].copy()
        s[0:3] += 0.02 * rng.normal(size=3)
        r = o.h_share_model(s, True)
        passes.append(dict(valid=True, h_x=r["h_x"], h=r["h"], R=r["R"]))
    passes[2] = dict(valid=False, h_x=np.zeros((0, o.C)), h=np.zeros(0), R=np.zeros(0))
    passes[3] = dict(valid=True, h_x=passes[3]["h_x"][:20], h=passes[3]["h"][:20], R=passes[3]["R"][:20])
    inp["upd_max_iter"] = np.array([max_iter], np.int32)
    inp["upd_valid"] = np.array([int(p["valid"]) for p in passes], np.int32)
    inp["upd_M"] = np.array([p["h_x"].shape[0] for p in passes], np.int32)
    for i, p in enumerate(passes):
        inp["upd_hx_%d" % i], inp["upd_h_%d" % i], inp["upd_R_%d" % i] = p["h_x"], p["h"], p["R"]
    inp["upd_state"], inp["upd_P"] = np.asarray(sc["state0"], np.float64), np.asarray(sc["P0"], np.float64)
    inp["upd_Rscalar"] = np.array([0.001], np.float64)
    # f-3: a 40-step predict chain
    steps = np.zeros((40, 7))
    steps[:, 0] = 0.005
    steps[:, 1:4] = np.array([0.1, -0.2, 9.8]) + 0.3 * rng.normal(size=(40, 3))
    steps[:, 4:7] = 0.2 * rng.normal(size=(40, 3))
    inp["pred_steps"] = steps
    Q = np.zeros((12, 12))
    Q[np.arange(12), np.arange(12)] = [1e-4] * 6 + [1e-5] * 6  # process_noise_cov(), use-ikfom.hpp:52-60
    inp["pred_Q"], inp["pred_state"], inp["pred_P"] = Q, inp["upd_state"].copy(), inp["upd_P"].copy()
    return inp, passes, sc


def oracle_outputs(inp, passes, sc):
    """The same quantities from the oracle restatement."""
    from oracle import orc
    out = {}
    K = inp["plane_pts"].shape[0]
    pab, pc, ok = np.zeros((K, 4), np.float32), np.zeros(K), np.zeros(K, np.int32)
    for k in range(K):
        n12 = np.zeros((5, 12), np.float32)
        n12[:, 0:3], n12[:, 5] = inp["plane_pts"][k, :, 0:3], inp["plane_pts"][k, :, 3]
        ok[k], pab[k], pc[k] = orc.esti_plane(n12, float(inp["plane_th"][0]), float(inp["cov_threshold"][0]))
    out["plane_pabcd"], out["plane_cov"], out["plane_ok"] = pab, pc, ok
    p12 = np.zeros((inp["unc_pts"].shape[0], 12), np.float32)
    p12[:, 0:3] = inp["unc_pts"]
    out["unc_cov"] = np.stack([orc.eval_point_uncertainty(p12[k], inp["unc_poses"][k]) for k in range(p12.shape[0])])
    out["comp_out"] = np.stack([orc.compound(a, b) for a, b in zip(inp["comp_a"], inp["comp_b"])])
    out["comp_inv_out"] = np.stack([orc.compound(a, b, inverse=True) for a, b in zip(inp["comp_a"], inp["comp_b"])])
    prm = dict(sc["params"])
    prm["max_iteration"] = int(inp["upd_max_iter"][0])
    o = orc.Oracle(prm, threads=1)
    o.set_replay(passes)
    u = o.update_iterated(inp["upd_state"], inp["upd_P"], R=float(inp["upd_Rscalar"][0]))
    out["upd_state_out"], out["upd_P_out"], out["upd_passes"] = u["state"], u["P"], np.array([u["passes"]], np.int32)
    x, P = inp["pred_state"].copy(), inp["pred_P"].copy()
    so, Po = [], []
    for r in inp["pred_steps"]:
        x, P = orc.predict(3, x, P, r[0], inp["pred_Q"], r[1:4], r[4:7])
        so.append(x.copy()), Po.append(P.copy())
    out["pred_state_out"], out["pred_P_out"] = np.stack(so), np.stack(Po)
    return out


def main():
    if not os.path.exists(REF_BIN):
        sys.exit("oracle/_ref/ref_eigen is not built (needs Eigen 3 + Boost + /root/reference): make -C oracle/ref_eigen")
    inp, _, _ = build_inputs()
    with tempfile.TemporaryDirectory() as d:
        fi, fo = os.path.join(d, "in.bin"), os.path.join(d, "out.bin")
        eigen_io.write(fi, inp)
        subprocess.check_call([REF_BIN, fi, fo])
        ref = eigen_io.read(fo)
    np.savez_compressed(GOLDEN, **{"in_" + k: v for k, v in inp.items()}, **{"ref_" + k: v for k, v in ref.items()})
    print("wrote", GOLDEN, "with", len(ref), "reference arrays")


if __name__ == "__main__":
    main()
