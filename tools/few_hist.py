import sys, os
sys.path.insert(0, os.getcwd())
import __graft_entry__ as ge; ge.load_package()
from malio_amd import capi, scenes
for cfg in (5, 2, 3):
    sc = scenes.make_scene(cfg=cfg)
    e = capi.Engine(sc["params"]); e.map_build(sc["map"]); e.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"])
    g = e.measure(sc["state0"], True)
    h1 = e.nfound_hist()
    e.scan_get()  # resolves the FEW points
    h2 = e.nfound_hist()
    print("cfg", cfg, "M", g["M"], "after search", h1, "after resolve", h2)
