"""Developer aid: what the bench contract's fences cost around a timed region - the first steps after a
torch.cuda.synchronize() and the closing synchronize itself (K = 20 regions run 0.6 us per step above K = 200 ones)."""
import sys, time
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as ge
ge.load_package()
from malio_amd import capi, scenes
sc = scenes.make_scene(cfg=2)
eng = capi.Engine(sc["params"], device=0)
eng.set_stream(torch.cuda.current_stream().cuda_stream)
eng.map_build(sc["map"]); eng.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"])
fast, out = eng.measure_fn(sc["state0"], True)
for _ in range(50): fast()
torch.cuda.synchronize()
ts = []
for rep in range(30):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); fast(); t1 = time.perf_counter(); fast(); t2 = time.perf_counter(); fast(); t3 = time.perf_counter()
    torch.cuda.synchronize(); t4 = time.perf_counter()
    ts.append(((t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6, (t4 - t3) * 1e6))
a = np.median(np.array(ts), axis=0)
print("first step after sync %.1f us, second %.1f, third %.1f, closing synchronize %.1f" % tuple(a))
