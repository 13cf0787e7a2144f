import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import __graft_entry__ as ge; ge.load_package()
from malio_amd import capi, scenes
sc = scenes.make_scene(cfg=int(os.environ.get("CFG", "2")))
eng = capi.Engine(sc["params"]); eng.map_build(sc["map"]); eng.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"])
eng.measure(sc["state0"], True)
print("counters", eng.debug_counters())
eng.set_profiling(True)
acc = {}
for k in range(20):
    eng.measure(sc["state0"], True)
    for n, ms in eng.last_kernel_times(): acc.setdefault(n, []).append(ms * 1000)
print("DBG", os.environ.get("MALIO_DBG"), {n: round(float(np.median(v)), 1) for n, v in acc.items()})
