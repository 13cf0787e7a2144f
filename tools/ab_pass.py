"""Developer aid: ONE library (MALIO_LIB or the shipped one), the bench's step at config CFG (full search pass, probe cache off):
wall time per pass (5 rounds of 300) and the median event time of its kernels; with UPD=1 also the gated update of a resident
scan. One line. Run it once per variant, interleaved (tools/ab_variants.sh)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge; ge.load_package()
from malio_amd import capi, scenes
cfg = int(os.environ.get("CFG", "2"))
sc = scenes.make_scene(cfg=cfg)
e = capi.Engine(sc["params"])
e.set_option("search_skip", 0).set_option("probe_cache", 0)
e.map_build(sc["map"]); e.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"])
fn, out = e.measure_fn(sc["state0"], True)
for _ in range(60): assert fn() >= 0
wall = []
for rnd in range(5):
    t = time.perf_counter()
    for _ in range(300): fn()
    wall.append((time.perf_counter() - t) / 300 * 1e6)
e.set_profiling(True)
acc = {}
for _ in range(40):
    fn()
    for n, ms in e.last_kernel_times(): acc.setdefault(n, []).append(ms * 1000)
e.set_profiling(False)
extra = ""
if os.environ.get("REUSE") == "1":
    fr, _ = e.measure_fn(sc["state0"], False)
    for _ in range(30): fr()
    t = time.perf_counter()
    for _ in range(300): fr()
    extra += " reuse_pass %.2f us" % ((time.perf_counter() - t) / 300 * 1e6)
    e.set_profiling(True)
    racc = {}
    for _ in range(30):
        fr()
        for n, ms in e.last_kernel_times(): racc.setdefault(n, []).append(ms * 1000)
    e.set_profiling(False)
    extra += " %s" % {n: round(float(np.median(v)), 1) for n, v in racc.items()}
if os.environ.get("UPD") == "1":
    e.set_option("probe_cache", 1)
    upd, res = e.update_iterated_fn(sc["state0"], sc["P0"])
    for _ in range(10): upd()
    t = time.perf_counter()
    for _ in range(100): upd()
    extra += " update_resident %.1f us (%d passes)" % ((time.perf_counter() - t) / 100 * 1e6, res()["passes"])
print("%s cfg%d pass %.2f us (%s) kernels %s M=%d%s" % (os.path.basename(os.environ.get("MALIO_LIB", "shipped")), cfg, float(np.median(wall)),
      " ".join("%.2f" % x for x in wall), {n: round(float(np.median(v)), 1) for n, v in acc.items()}, out.M, extra))
