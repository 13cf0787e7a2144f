"""Wall time of the map mutators at BASELINE configs[1] scale (1M-point map, one scan's additions)."""
import sys, time
import numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import __graft_entry__ as ge
ge.load_package()
from malio_amd import capi, scenes

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
sc = scenes.make_scene(cfg=cfg)
e = capi.Engine(sc["params"])
t = time.perf_counter(); e.map_build(sc["map"]); print("map_build %d pts: %.2f ms" % (sc["Nmap"], (time.perf_counter() - t) * 1e3))
t = time.perf_counter(); e.map_build(sc["map"]); print("map_build again: %.2f ms" % ((time.perf_counter() - t) * 1e3))
e.nearest_search(sc["map"][:8], 5)  # first use of the batched search (module load) outside the timings
rng = np.random.default_rng(0)
for n in (10000, 10000, 50000):
    new = sc["map"][rng.integers(0, sc["Nmap"], n)].copy()
    new[:, :3] += rng.normal(0, 0.3, size=(n, 3)).astype(np.float32)
    t = time.perf_counter(); a = e.map_add(new, True); dt = time.perf_counter() - t
    print("map_add(ds) %d pts -> %d insertions, size %d: %.2f ms" % (n, a, e.map_size(), dt * 1e3))
    t = time.perf_counter(); e.map_add(new[:n // 10], False); dt = time.perf_counter() - t
    print("map_add(no ds) %d pts, size %d: %.2f ms" % (n // 10, e.map_size(), dt * 1e3))
    t = time.perf_counter(); e.nearest_search(new[:8], 5); dt = time.perf_counter() - t
    print("   next search (rebuilds the lists only if the in-place update did not fit): %.2f ms" % (dt * 1e3))
    t = time.perf_counter(); e.nearest_search(new[:8], 5); dt = time.perf_counter() - t
    print("   search again: %.2f ms" % (dt * 1e3))
box = np.array([[0, 0, -5, 30, 30, 5]], np.float32)
t = time.perf_counter(); d = e.map_delete_boxes(box); dt = time.perf_counter() - t
print("delete_boxes -> %d deleted, size %d: %.2f ms" % (d, e.map_size(), dt * 1e3))
t = time.perf_counter(); e.nearest_search(sc["map"][:8], 5); dt = time.perf_counter() - t
print("   next search: %.2f ms" % (dt * 1e3))
