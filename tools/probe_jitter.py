"""developer aid: distribution of update_iterated wall times per update mode (outliers)"""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import __graft_entry__ as ge; ge.load_package()
from malio_amd import capi, scenes
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
sc = scenes.make_scene(cfg=cfg)
for mode in ("host", "gated", "host", "gated"):
    eng = capi.Engine(sc["params"]); eng.set_update_mode(mode); eng.map_build(sc["map"])
    ts = []
    for rep in range(reps):
        eng.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"])
        eng.measure(sc["state0"], True)
        t = time.perf_counter(); u = eng.update_iterated(sc["state0"], sc["P0"]); ts.append(time.perf_counter() - t)
    ts = np.array(ts[5:]) * 1e6
    print(mode, "p10 %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f  n>1ms %d" % (*np.percentile(ts, [10, 50, 90, 99]), ts.max(), int((ts > 1000).sum())), flush=True)
