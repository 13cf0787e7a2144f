import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import __graft_entry__ as ge; ge.load_package()
import torch
from malio_amd import capi, scenes, dist as mdist
from oracle import orc
sc = scenes.make_scene(seed=221, N=3000, Nmap=40000, L=3)
eng = capi.Engine(sc["params"]); eng.map_build(sc["map"]); eng.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"])
o = orc.Oracle(sc["params"], threads=4); o.map_build(sc["map"]); o.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"])
w = o.update_iterated(sc["state0"], sc["P0"]); print("oracle", w["passes"], w["searches"], w["M"])
v = eng.update_iterated(sc["state0"], sc["P0"]); print("eng", v["passes"], v["searches"], v["M"], v["t"])
be = mdist.HipBackend(eng)
eng.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"])
u = mdist.sharded_update_iterated(be, sc["state0"], sc["P0"]); print("sharded", u["passes"], u["searches"], u["M"], u["t"])
print(np.abs(u["state"]-w["state"]).max(), np.abs(v["state"]-w["state"]).max())
