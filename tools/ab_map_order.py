"""Round 6: what the order of the MAP ARRAY costs the search pass. The scene generator leaves the map in raster order (the
gather of a query's five neighbours then shares lines with its neighbours' gathers); a map that grew scan by scan does not.
Four engines in one process, same scan: map handed over in raster / shuffled order x MALIO_OPT_MAP_CELL_ORDER off / on; per
engine the wall time of the bench's step (full search pass, probe cache off) and the median event time of its kernels;
rounds interleaved. Results must be the same point sets: M and the sums (to summation order) are compared.
   CFG=2 python tools/ab_map_order.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge; ge.load_package()
from malio_amd import capi, scenes
cfg = int(os.environ.get("CFG", "2"))
sc = scenes.make_scene(cfg=cfg)
shuf = sc["map"][np.random.default_rng(77).permutation(sc["Nmap"])]
engs = []
for mname, mp in (("raster", sc["map"]), ("shuffled", shuf)):
    for co in (0, 1):
        e = capi.Engine(sc["params"])
        e.set_option("map_cell_order", co).set_option("search_skip", 0).set_option("probe_cache", 0)
        t0 = time.perf_counter(); e.map_build(mp); tb = time.perf_counter() - t0
        e.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"])
        fn, out = e.measure_fn(sc["state0"], True)
        for _ in range(50): assert fn() >= 0
        engs.append(dict(name="%s, cell order %d" % (mname, co), e=e, fn=fn, out=out, wall=[], build_ms=tb * 1e3))
for rnd in range(5):
    for g in engs:
        t = time.perf_counter()
        for _ in range(300): g["fn"]()
        g["wall"].append((time.perf_counter() - t) / 300 * 1e6)
ref = None
for g in engs:
    e = g["e"]
    e.set_profiling(True)
    acc = {}
    for _ in range(30):
        g["fn"]()
        for n, ms in e.last_kernel_times(): acc.setdefault(n, []).append(ms * 1000)
    e.set_profiling(False)
    H = np.array(g["out"].HtRinvH[:])
    if ref is None: ref = (g["out"].M, H)
    assert g["out"].M == ref[0] and np.abs(H - ref[1]).max() <= 1e-10 * np.abs(ref[1]).max(), g["name"]
    print("%-24s pass %.2f us (rounds %s)  kernels %s  map_build %.1f ms  M=%d" % (
        g["name"], float(np.median(g["wall"])), " ".join("%.2f" % x for x in g["wall"]),
        {n: round(float(np.median(v)), 1) for n, v in acc.items()}, g["build_ms"], g["out"].M))
