"""Developer aid: ONE shard's node pass (malio_measure_node over a one-rank local exchange) timed alone on this GPU - what a
GPU of a G-GPU node spends per pass before the exchange itself. CFG (default 4), G (default 8), MALIO_FUSE=0 for the
three-kernel form."""
import os, sys, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge; ge.load_package()
from malio_amd import capi, scenes
cfg, G = int(os.environ.get("CFG", "4")), int(os.environ.get("G", "8"))
sc = scenes.make_scene(cfg=cfg)
N, L = sc["N"], sc["L"]
row = L * 97 + 8
class X:  # a one-rank local exchange
    def __init__(self):
        self.arr = (C.c_void_p * 1)()
        assert capi.lib().malio_xchg_create_local(1, row, self.arr) == 0
        self.h = C.c_void_p(self.arr[0])
for part in ("tiles", "scan"):
    e = capi.Engine(sc["params"]); e.set_option("search_skip", 0)
    if part == "tiles":
        e.set_partition(0, G, float(os.environ.get("TILE", "16")))
        scan = sc["scan"]
    else:
        scan = sc["scan"][: N // G]
    e.map_build(sc["map"])
    e.scan_set(scan, sc["tables"], sc["temporal_comp"])
    x = X()
    fn, out = e.measure_node_fn(x, sc["state0"], True)
    fr, _ = e.measure_node_fn(x, sc["state0"], False)
    for f, name in ((fn, "search"), (fr, "reuse")):
        for _ in range(30): assert f() >= 0
        ts = []
        for _ in range(300):
            t = time.perf_counter(); f(); ts.append(time.perf_counter() - t)
        print("%s shard 0 of %d, %s pass: %.1f us (M %d)  fuse %s" % (part, G, name, np.median(ts) * 1e6, out.M, e.fuse_stats()))
    e.set_profiling(True)
    acc = {}
    for _ in range(20):
        fn()
        for n, ms in e.last_kernel_times(): acc.setdefault(n, []).append(ms * 1000)
    print("   kernels (events):", {n: round(float(np.median(v)), 1) for n, v in acc.items()})
    e.set_profiling(False)
