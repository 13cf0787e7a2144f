"""developer aid: wall time of malio_scan_set for a pageable and a page-locked cloud (pre-built ctypes arguments)"""
import sys, time, os
import ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import __graft_entry__ as ge
ge.load_package()
from malio_amd import capi, scenes
sc = scenes.make_scene(cfg=2)
e = capi.Engine(sc["params"]); e.map_build(sc["map"])
L = e.L
tabs = [np.ascontiguousarray(np.asarray(t, np.float64).reshape(-1, 59)) for t in sc["tables"]]
ptrs = (C.POINTER(capi.Pose) * L)(*[capi._p(t, capi.Pose) for t in tabs])
lens = (C.c_int * L)(*[t.shape[0] for t in tabs])
tc = np.ascontiguousarray(np.asarray(sc["temporal_comp"], np.float64).reshape(-1, 59))
tcp = capi._p(tc, capi.Pose)
for n in (10000, 100000):
    pts = np.ascontiguousarray(sc["scan"][:n])
    pin = capi.PinnedArray(pts.shape, np.float32); pin.array[:] = pts
    for name, arr in (("pageable", pts), ("pinned", pin.array)):
        ts = []
        for k in range(10):
            e.N = n
            p = capi._p(arr, capi.Point)
            t = time.perf_counter(); rc = capi.lib().malio_scan_set(e.h, p, n, ptrs, lens, tcp); ts.append(time.perf_counter() - t)
            assert rc == 0
            t = time.perf_counter(); e.measure(sc["state0"], True); t1 = time.perf_counter() - t
        print("n=%6d %-8s scan_set %.1f us (min %.1f)   first pass after it %.1f us" % (n, name, np.median(ts[2:]) * 1e6, min(ts) * 1e6, t1 * 1e6), flush=True)
