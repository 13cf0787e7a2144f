import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import __graft_entry__ as ge; ge.load_package()
from malio_amd import capi, scenes
sc = scenes.make_scene(cfg=int(os.environ.get("CFG", "2")))
eng = capi.Engine(sc["params"]); eng.map_build(sc["map"]); eng.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"])
eng.measure(sc["state0"], True); eng.debug_counters()
eng.measure(sc["state0"], True)
print("pending per pass", eng.debug_counters())
