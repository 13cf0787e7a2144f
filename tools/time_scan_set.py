import sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import __graft_entry__ as ge
ge.load_package()
from malio_amd import capi, scenes
sc = scenes.make_scene(cfg=2)
e = capi.Engine(sc["params"]); e.map_build(sc["map"])
for n in (1000, 10000, 50000, 100000):
    pts = np.ascontiguousarray(sc["scan"][:n])
    ts = []
    for k in range(8):
        t = time.perf_counter(); e.scan_set(pts, sc["tables"], sc["temporal_comp"]); ts.append(time.perf_counter() - t)
        e.measure(sc["state0"], True)
    print("n=%6d scan_set %.1f us (min %.1f)" % (n, np.median(ts[2:]) * 1e6, min(ts) * 1e6))
