"""Developer aid: k_search time vs scan size around the one-generation limit (256 CUs x 6 workgroups x 64 queries)."""
import sys, time
import numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import __graft_entry__ as ge
ge.load_package()
from malio_amd import capi, scenes
sc = scenes.make_scene(cfg=2)
e = capi.Engine(sc["params"]); e.map_build(sc["map"])
for n in [int(a) for a in sys.argv[1:]] or (90000, 96000, 98304, 98368, 99000, 100000):
    e.scan_set(sc["scan"][:n], sc["tables"], sc["temporal_comp"])
    fn, out = e.measure_fn(sc["state0"], True)
    for _ in range(30): fn()
    t = time.perf_counter()
    for _ in range(300): fn()
    dt = (time.perf_counter() - t) / 300
    e.set_profiling(True)
    per = {}
    for _ in range(40):
        e.measure(sc["state0"], True)
        for name, ms in e.last_kernel_times(): per.setdefault(name, []).append(ms)
    e.set_profiling(False)
    print("N=%6d blocks=%4d  %.2f us/pass  %s" % (n, (n + 63) // 64, dt * 1e6, {k: round(float(np.median(v)) * 1e3, 1) for k, v in per.items()}))
