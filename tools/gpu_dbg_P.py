import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import __graft_entry__ as ge; ge.load_package()
from malio_amd import capi, scenes
from oracle import orc
np.set_printoptions(linewidth=200, precision=3)
sc = scenes.make_scene(cfg=2)
eng = capi.Engine(sc["params"]); eng.map_build(sc["map"]); eng.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"])
o = orc.Oracle(sc["params"], threads=16, use_ref=True); o.map_build(sc["map"]); o.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"])
u = eng.update_iterated(sc["state0"], sc["P0"]); v = o.update_iterated(sc["state0"], sc["P0"])
print("passes", u["passes"], v["passes"], u["M"], v["M"])
d = u["state"] - v["state"]; print("state diff", d)
P, Q = u["P"], v["P"]; dg = np.sqrt(np.diag(Q))
print("corr diff blocks (6x6 top):"); print((np.abs(P-Q)/np.outer(dg,dg))[:9,:9])
# oracle trace of states
print("trace oracle:"); print(v["trace"][:, :7])
# GPU: replay passes manually with oracle states to compare sums at the final state
xf = v["trace"][-2] if len(v["trace"])>1 else sc["state0"]
g = eng.measure(xf, True); r = o.h_share_model(xf, True)
Rc = np.where(r["R"]<1e-4,1e-3,r["R"]); HtH=(r["h_x"].T/Rc)@r["h_x"]
print("at final-pass state: M", g["M"], r["M"], "HtH rel", np.abs(g["HtRinvH"]-HtH).max()/np.abs(HtH).max())
