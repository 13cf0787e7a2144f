"""How often do two map points lie at EXACTLY the same float distance from a query, and how does the reference order them?
CPU only (the reference ikd-Tree compiled under oracle/_ref): K search passes of BASELINE config CFG at iterates a few
centimetres apart; every query whose five neighbours contain two equal d2 (ikd_Tree.cpp:1697's float arithmetic) is listed with
the map indices of the tied pair in the order Nearest_Search returned them. The engine orders such pairs by (d2, slot);
tests/conftest.py::exact_ties is what the parity tests do about them.     python tools/tie_census.py [cfg=5] [passes=12]"""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge; ge.load_package()
from malio_amd import scenes
from oracle import orc
from scipy.spatial import cKDTree
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 5
K = int(sys.argv[2]) if len(sys.argv) > 2 else 12
sc = scenes.make_scene(cfg=cfg)
o = orc.Oracle(sc["params"], threads=8, use_ref=True)
o.map_build(sc["map"]); o.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"])
t = cKDTree(sc["map"][:, :3])
rng = np.random.default_rng(3)
asc = desc = nq = 0
for k in range(K):
    s = sc["state0"].copy(); s[0:3] += rng.normal(0, 0.02, 3)
    s[3:7] = scenes.q_norm(scenes.q_mul(s[3:7], scenes.q_from_rotvec(rng.normal(0, 0.002, 3))))
    o.h_share_model(s, True); g = o.scan_get()
    near = g["nearest"][:, :, :3].astype(np.float32); w = g["world"].astype(np.float32)
    d = w[:, None, :] - near
    d2 = (d[:, :, 0] * d[:, :, 0] + d[:, :, 1] * d[:, :, 1]) + d[:, :, 2] * d[:, :, 2]
    nq += len(w)
    for i in np.nonzero((d2[:, 1:] == d2[:, :-1]).any(1))[0]:
        _, j = t.query(near[i], k=1)
        for a in range(4):
            if d2[i, a] == d2[i, a + 1] and j[a] != j[a + 1]:
                asc += j[a] < j[a + 1]; desc += j[a] > j[a + 1]
                print("pass %2d query %6d  d2 %-10.7g map indices in the reference's order: %7d %7d  accepted: %d" % (k, i, d2[i, a], j[a], j[a + 1], g["selected"][i]))
print("config %d: %d queries, %d tied pairs inside the five neighbours; the reference returned %d in ascending map index, %d in descending" % (cfg, nq, asc + desc, asc, desc))
