"""Developer aid: wall time of malio_map_incremental over consecutive scans (config 2 geometry)."""
import sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import __graft_entry__ as ge
ge.load_package()
from malio_amd import capi, scenes
import os
cfg = int(os.environ.get("CFG", "2"))
sc = scenes.make_scene(cfg=cfg)
e = capi.Engine(sc["params"]); e.map_build(sc["map"])
wny = np.full(sc["N"], 0.001, np.float32)
for k in range(6):
    s2 = scenes.make_scene(cfg=cfg, scan_seed=100 + k)
    t = time.perf_counter(); e.scan_set(s2["scan"], sc["tables"], sc["temporal_comp"]); t_set = time.perf_counter() - t
    t = time.perf_counter(); u = e.update_iterated(sc["state0"], sc["P0"]); t_up = time.perf_counter() - t
    t = time.perf_counter(); na, nn, ret = e.map_incremental(u["state"], True, wny); t_inc = time.perf_counter() - t
    print("scan %d: scan_set %.2f ms  update %.2f ms (%d passes)  map_incremental %.2f ms (PointToAdd %d, NoNeedDownsample %d)  %s" % (
        k, t_set * 1e3, t_up * 1e3, u["passes"], t_inc * 1e3, na, nn, {k2: v for k2, v in e.debug_counters().items() if k2 in ("rebuilds", "inplace", "tombstones", "map_n", "dead", "nl1_cells")}))
