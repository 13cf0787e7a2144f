"""Developer aid: wall time of the bench's step (malio_measure, full search pass, config CFG) with ONE option at two values,
interleaved on two handles of one process, 5 rounds of 300 steps:  python tools/ab_opt.py done_stamps 0 1"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge; ge.load_package()
from malio_amd import capi, scenes
name, va, vb = sys.argv[1], float(sys.argv[2]), float(sys.argv[3])
sc = scenes.make_scene(cfg=int(os.environ.get("CFG", "2")))
fns = []
for v in (va, vb):
    e = capi.Engine(sc["params"]); e.set_option(name, v); e.set_option("search_skip", 0)
    e.map_build(sc["map"]); e.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"])
    fn, out = e.measure_fn(sc["state0"], True)
    for _ in range(50): assert fn() >= 0
    fns.append((v, fn, e, out))
res = {va: [], vb: []}
for rnd in range(5):
    for v, fn, e, out in fns:
        t = time.perf_counter()
        for _ in range(300): fn()
        res[v].append((time.perf_counter() - t) / 300 * 1e6)
for v in (va, vb):
    print("%s = %g: %.2f us per pass (rounds %s)  M=%d" % (name, v, float(np.median(res[v])), " ".join("%.2f" % x for x in res[v]), [o for vv, f, e, o in fns if vv == v][0].M))
