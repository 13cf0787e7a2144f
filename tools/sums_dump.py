"""Developer aid: the 97 sums of a search pass and a reuse pass (fused and three-kernel path, configs 1, 2, 5) into an .npz - run it
under two builds (MALIO_LIB=.../variants/a.so, b.so) and compare the files bit for bit: how a change of the summation code (the
DPP butterflies of k_final_reduce, round 5) is shown to leave every bit alone.   python tools/sums_dump.py out.npz"""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge; ge.load_package()
from malio_amd import capi, scenes
out = {}
for cfg in (1, 2, 5):
    sc = scenes.make_scene(cfg=cfg)
    for fuse in (1, 0):
        e = capi.Engine(sc["params"]); e.set_option("fuse", fuse)
        e.map_build(sc["map"]); e.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"])
        for k in range(3):
            m = e.measure(sc["state0"], True)
        m2 = e.measure(sc["state0"], False)
        out["c%d_f%d_s" % (cfg, fuse)] = m["HtRinvH"]; out["c%d_f%d_r" % (cfg, fuse)] = m2["HtRinvH"]
        out["c%d_f%d_z" % (cfg, fuse)] = m["HtRinvz"] if "HtRinvz" in m else np.zeros(1)
np.savez(sys.argv[1], **out)
