import os, sys, time
import numpy as np, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge; ge.load_package()
from malio_amd import capi, scenes
CFG = int(os.environ.get("CFG", "2"))
sc = scenes.make_scene(cfg=CFG)
e = capi.Engine(sc["params"]); e.map_build(sc["map"])
upd, upd_result = e.update_iterated_fn(sc["state0"], sc["P0"])
minc, cnt = e.map_incremental_fn(None, True)
T = 8
PACKED = os.environ.get("POINTS") != "1"
bufs = [capi.PinnedArray((sc["N"], 5 if PACKED else 12), np.float32) for _ in range(T + 1)]
for k in range(T + 1):
    a = scenes.make_scene(cfg=CFG, scan_seed=700 + k)["scan"]
    bufs[k].array[:] = capi.Engine.pack_scan(a) if PACKED else a
calls = [(e.scan_set_packed_fn if PACKED else e.scan_set_fn)(b.array, sc["tables"], sc["temporal_comp"]) for b in bufs]
stage = capi.lib().malio_scan_stage
ptrs = [C.c_void_p(b.array.ctypes.data) for b in bufs]
import torch
for rep in range(int(os.environ.get("REPS", "3"))):
    calls[0](); assert upd() == 0
    st = capi.state_from_flat(upd_result()["state"], sc["L"])
    torch.cuda.synchronize()
    ts = {"stage": 0, "minc": 0, "set": 0, "upd": 0}
    t0 = time.perf_counter()
    for k in range(1, T + 1):
        a = time.perf_counter(); stage(e.h, ptrs[k], sc["N"], 1 if PACKED else 0)
        b = time.perf_counter(); minc(st)
        c_ = time.perf_counter(); calls[k]()
        d = time.perf_counter(); assert upd() == 0
        f = time.perf_counter()
        ts["stage"] += b - a; ts["minc"] += c_ - b; ts["set"] += d - c_; ts["upd"] += f - d
    torch.cuda.synchronize()
    tot = (time.perf_counter() - t0) / T * 1e6
    minc(st)
    print("turn %.1f us | " % tot + "  ".join("%s %.1f" % (k, v / T * 1e6) for k, v in ts.items()))
print("fuse stats", e.fuse_stats(), {k: v for k, v in e.debug_counters().items() if k in ("rebuilds", "inplace")})
