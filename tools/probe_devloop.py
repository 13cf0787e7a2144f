"""developer aid: device-resident update loop vs the host-driven loop on one box - agreement and wall time"""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import __graft_entry__ as ge; ge.load_package()
from malio_amd import capi, scenes
for cfg in [int(a) for a in (sys.argv[1:] or ["2"])]:
    sc = scenes.make_scene(cfg=cfg)
    res = {}
    for mode in ("host", "gated", "device"):
        eng = capi.Engine(sc["params"]); eng.set_update_mode(mode); eng.map_build(sc["map"])
        ts = []
        for rep in range(12):
            eng.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"])
            eng.measure(sc["state0"], True)
            t = time.perf_counter(); u = eng.update_iterated(sc["state0"], sc["P0"]); ts.append(time.perf_counter() - t)
        res[mode] = u
        if mode == "gated":
            import ctypes as C
            tr = (C.c_double * 60)()
            ntr = capi.lib().malio_debug_gate_trace(eng.h, tr)
            print("   all reps [us]:", [round(x * 1e6) for x in ts])
            print("   gated trace (loop top, launched, pre done, sums seen, published) per pass:", [[round(tr[k + j], 1) for j in range(5)] for k in range(0, ntr - 4, 5)], flush=True)
        print("cfg", cfg, mode, "update_ms median %.4f min %.4f" % (np.median(ts[2:]) * 1e3, min(ts) * 1e3), "passes", u["passes"], "searches", u["searches"], "M", u["M"], "t", u["t"], flush=True)
    import ctypes as C
    st = (C.c_longlong * 16)()
    if capi.lib().malio_debug_loop_stamps(eng.h, st) == 0:
        v = list(st)
        print("   last step kernel, us since entry:", [round((x - v[0]) / 100.0, 1) for x in v[:11]], flush=True)
    for a_, b_ in (("device", "host"), ("gated", "host")):
        u, v = res[a_], res[b_]
        print("  ", a_, "vs", b_, end=": ")
        dg = np.sqrt(np.abs(np.diag(v["P"])))
        print("   |dstate| max %.3e   |dP| corr-scale max %.3e   |dP|/|P|max %.3e" % (
            np.abs(u["state"] - v["state"]).max(), (np.abs(u["P"] - v["P"]) / (np.outer(dg, dg) + 1e-300)).max(),
            np.abs(u["P"] - v["P"]).max() / np.abs(v["P"]).max()), flush=True)
