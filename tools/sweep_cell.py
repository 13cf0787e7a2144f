"""Developer aid: search-pass time vs level-1 cell edge (config 2)."""
import sys, time
import numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import __graft_entry__ as ge
ge.load_package()
from malio_amd import capi, scenes
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
sc = scenes.make_scene(cfg=cfg)
for cs in [float(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else "0.875,0.9375,1.0,1.0625,1.125,1.25".split(","))]:
    e = capi.Engine(sc["params"], cell_size=cs)
    e.map_build(sc["map"]); e.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"])
    fn, out = e.measure_fn(sc["state0"], True)
    for _ in range(30): fn()
    t = time.perf_counter()
    for _ in range(300): fn()
    dt = (time.perf_counter() - t) / 300
    e.set_profiling(True)
    per = {}
    for _ in range(30):
        e.measure(sc["state0"], True)
        for n, ms in e.last_kernel_times(): per.setdefault(n, []).append(ms)
    e.set_profiling(False)
    print("cell %.3f: %.2f us/pass  M=%d  kernels %s  dbg %s" % (cs, dt * 1e6, out.M, {k: round(float(np.mean(v)) * 1e3, 1) for k, v in per.items()}, e.debug_counters()))
    del e
