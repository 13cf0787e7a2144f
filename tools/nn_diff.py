"""Developer aid: one search pass of BASELINE config CFG (default 5) at an iterate a centimetre from the prior, engine against
oracle (reference ikd-Tree inside); prints every query whose Nearest_Points differ, with both sides distances - how round 6 found
that the two sides order exact ties of float distances differently (tests/conftest.py::exact_ties). CO=1,0: MALIO_OPT_MAP_CELL_ORDER."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge; ge.load_package()
from malio_amd import capi, scenes
from oracle import orc
cfg = int(os.environ.get("CFG", "5"))
sc = scenes.make_scene(cfg=cfg)
def move(s0, dp, dr):
    s = s0.copy(); s[0:3] += dp
    s[3:7] = scenes.q_norm(scenes.q_mul(s[3:7], scenes.q_from_rotvec(dr))); return s
st = move(sc["state0"], [0.012, -0.02, 0.006], [0.001, -0.002, 0.0015])
o = orc.Oracle(sc["params"], threads=16, use_ref=True)
o.map_build(sc["map"]); o.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"])
o.h_share_model(st, True); og = o.scan_get()
for co in (int(x) for x in os.environ.get("CO", "1,0").split(",")):
    e = capi.Engine(sc["params"]); e.set_option("map_cell_order", co)
    e.map_build(sc["map"]); e.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"])
    e.measure(sc["state0"], True); e.measure(st, True)
    g = e.scan_get()
    bad = np.nonzero((g["nearest"][:, :, :3] != og["nearest"][:, :, :3]).any((1, 2)))[0]
    print("cell order", co, "queries whose Nearest_Points differ:", len(bad), "cnt equal:", np.array_equal(g["nearest_cnt"], og["nearest_cnt"]))
    for i in bad[:6]:
        w = g["world"][i]
        def d2(n):
            d = w[None, :] - n[:, :3]
            return (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
        print(" query", i, "world", w, "sel gpu/orc", g["selected"][i], og["selected"][i], "cnt", g["nearest_cnt"][i])
        print("   gpu d2", d2(g["nearest"][i]).tolist())
        print("   orc d2", d2(og["nearest"][i]).tolist())
        print("   gpu pts", g["nearest"][i][:, :3].tolist())
        print("   orc pts", og["nearest"][i][:, :3].tolist())
