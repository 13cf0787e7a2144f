"""Developer aid: consecutive turns of the mapping loop (scan_set -> update_iterated -> map_incremental, config 2
geometry) the way the integration drives it: the cloud in page-locked memory, the C calls with their arguments
marshalled beforehand, world_normal_y = NULL. PAGEABLE=1 / WNY=1 select the slower variants. Run it under
rocprofv3 --kernel-trace --memory-copy-trace for tools/loop_timeline.py."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
ge.load_package()
from malio_amd import capi, scenes
cfg = int(os.environ.get("CFG", "2"))
sc = scenes.make_scene(cfg=cfg)
e = capi.Engine(sc["params"]); e.map_build(sc["map"])
pin = capi.PinnedArray(sc["scan"].shape, np.float32)
upd, upd_result = e.update_iterated_fn(sc["state0"], sc["P0"])
minc, cnt = e.map_incremental_fn(np.full(sc["N"], 0.001, np.float32) if os.environ.get("WNY") == "1" else None, True)
for k in range(6):
    s2 = scenes.make_scene(cfg=cfg, scan_seed=100 + k)
    if os.environ.get("PAGEABLE") == "1":
        src = s2["scan"]
    else:
        pin.array[:] = s2["scan"]
        src = pin.array
    call = e.scan_set_fn(src, sc["tables"], sc["temporal_comp"])
    time.sleep(0.002)  # (whatever the previous turn left queued is done: every turn starts from an idle GPU)
    t = time.perf_counter(); call(); t_set = time.perf_counter() - t
    t = time.perf_counter(); assert upd() == 0; t_up = time.perf_counter() - t
    u = upd_result()
    st = capi.state_from_flat(u["state"], sc["L"])
    t = time.perf_counter(); minc(st); t_inc = time.perf_counter() - t
    print("scan %d: scan_set %.3f ms  update %.3f ms (%d passes)  map_incremental %.3f ms (PointToAdd %d, NoNeedDownsample %d)" % (
        k, t_set * 1e3, t_up * 1e3, u["passes"], t_inc * 1e3, cnt[0], cnt[1]))
print({k2: v for k2, v in e.debug_counters().items() if k2 in ("rebuilds", "inplace", "tombstones", "map_n", "dead", "nl1_cells")})
