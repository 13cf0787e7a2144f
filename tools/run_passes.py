"""Developer aid for kernel traces: N search passes, N reuse passes, N gated updates on config CFG (default 2)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge; ge.load_package()
from malio_amd import capi, scenes
sc = scenes.make_scene(cfg=int(os.environ.get("CFG", "2")))
e = capi.Engine(sc["params"]); e.map_build(sc["map"]); e.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"])
fs, _ = e.measure_fn(sc["state0"], True); fr, _ = e.measure_fn(sc["state0"], False)
for _ in range(300): fs()
for _ in range(300): fr()
upd, res = e.update_iterated_fn(sc["state0"], sc["P0"])
for _ in range(60):
    e.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"]); e.measure(sc["state0"], True); upd()
print(e.fuse_stats(), res()["passes"])
