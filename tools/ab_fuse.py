"""Developer aid: one-kernel pass (k_pass) against the three-kernel pass, two handles of one process on one box,
interleaved: wall time of a search pass / a reuse pass / the whole gated update, and the kernel event times.
    CFG=2 python tools/ab_fuse.py"""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import __graft_entry__ as ge; ge.load_package()
from malio_amd import capi, scenes
cfg = int(os.environ.get("CFG", "2"))
sc = scenes.make_scene(cfg=cfg)
engs = {}
for name, env in (("three", "0"), ("one", "1")):
    os.environ["MALIO_FUSE"] = env
    e = capi.Engine(sc["params"]); e.map_build(sc["map"]); e.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"])
    e.measure(sc["state0"], True); e.measure(sc["state0"], True)  # (reads the environment at its first eligible pass)
    engs[name] = e
def wall(fn, n=300):
    for _ in range(20): fn()
    ts = []
    for _ in range(n):
        t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
    return np.median(ts) * 1e6, np.percentile(ts, 90) * 1e6
res = {}
for rep in range(3):
    for name, e in engs.items():
        fs, _ = e.measure_fn(sc["state0"], True)
        fr, _ = e.measure_fn(sc["state0"], False)
        res.setdefault(name + " search", []).append(wall(fs)[0])
        res.setdefault(name + " reuse", []).append(wall(fr)[0])
        upd, result = e.update_iterated_fn(sc["state0"], sc["P0"])
        ts = []
        for k in range(40):
            e.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"]); e.measure(sc["state0"], True)
            t = time.perf_counter(); rc = upd(); ts.append(time.perf_counter() - t)
            assert rc == 0
        res.setdefault(name + " update", []).append(np.median(ts[5:]) * 1e6)
        r = result()
        res.setdefault(name + " passes", []).append(r["passes"])
for k, v in res.items():
    print("%-16s %s" % (k, " ".join("%7.1f" % x for x in v)))
for name, e in engs.items():
    print(name, "fuse stats", e.fuse_stats())
    e.set_profiling(True)
    for conv in (True, False):
        acc = {}
        for k in range(20):
            e.measure(sc["state0"], conv)
            for n, ms in e.last_kernel_times(): acc.setdefault(n, []).append(ms * 1000)
        print(name, "search" if conv else "reuse", "KERNELS", {n: round(float(np.median(v)), 1) for n, v in acc.items()})
    e.set_profiling(False)
