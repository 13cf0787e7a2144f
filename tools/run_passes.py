"""Developer aid for kernel traces: N full search passes, N search passes that keep cached neighbours (two iterates 1.5 cm
apart in turn), N reuse passes, N gated updates of a NEW scan each on config CFG (default 2)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge; ge.load_package()
from malio_amd import capi, scenes
cfg = int(os.environ.get("CFG", "2"))
sc = scenes.make_scene(cfg=cfg)
e = capi.Engine(sc["params"]); e.map_build(sc["map"]); e.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"])
s2 = sc["state0"].copy(); s2[0:3] += [0.01, -0.008, 0.004]
fs, _ = e.measure_fn(sc["state0"], True); fs2, _ = e.measure_fn(s2, True); fr, _ = e.measure_fn(sc["state0"], False)
e.set_option("search_skip", 0)
e.set_option("probe_cache", 0)   # the bench's headline step: every point probes the directory
for _ in range(300): fs()
e.set_option("probe_cache", 1)   # (library default from here on)
e.set_option("search_skip", 1)
for k in range(300): (fs2 if k & 1 else fs)()
print("skip", e.skip_stats())
for _ in range(300): fr()
upd, res = e.update_iterated_fn(sc["state0"], sc["P0"])
scans = [scenes.make_scene(cfg=cfg, scan_seed=950 + k)["scan"] for k in range(4)]
for k in range(60):
    e.scan_set(scans[k % 4], sc["tables"], sc["temporal_comp"]); upd()
print(e.fuse_stats(), res()["passes"], e.skip_stats())
