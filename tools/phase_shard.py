"""Developer aid (variant built with -DMALIO_PHASE_CLOCK): entry / exit times of the workgroups of k_pass on ONE tile shard of G
(CFG default 4, G default 8): MALIO_LIB=.../variants/phase.so python tools/phase_shard.py"""
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge; ge.load_package()
from malio_amd import capi, scenes
cfg, G = int(os.environ.get("CFG", "4")), int(os.environ.get("G", "8"))
sc = scenes.make_scene(cfg=cfg)
N, L = sc["N"], sc["L"]
for part in ("tiles", "scan"):
    e = capi.Engine(sc["params"])
    if part == "tiles":
        e.set_partition(0, G, 16.0)
        scan = sc["scan"]
    else:
        scan = sc["scan"][: N // G]
    e.map_build(sc["map"]); e.scan_set(scan, sc["tables"], sc["temporal_comp"])
    for _ in range(10):
        e.measure(sc["state0"], True)
    nb = min(8192, (scan.shape[0] + 63) // 64 + 4)
    sp = (C.c_longlong * (4 * nb))()
    assert capi.lib().malio_debug_span(sp, nb) == 0
    sp = np.array(sp[:], np.int64).reshape(4, nb)
    live = sp[0] > 0
    t0 = sp[0][live].min()
    ent = (sp[0][live] - t0) / 100.0
    own = live & (sp[1] > sp[0])
    ex = (sp[1][own] - t0) / 100.0
    print("%s shard: %d workgroups entered (entry median %.2f, 90%% %.2f, last %.2f us); %d owned, first owned index %d last %d; owned exit first %.2f median %.2f last %.2f us; owned residence median %.2f" % (
        part, live.sum(), np.median(ent), np.percentile(ent, 90), ent.max(), own.sum(), np.where(own)[0].min(), np.where(own)[0].max(), ex.min(), np.median(ex), ex.max(),
        np.median((sp[1][own] - sp[0][own]) / 100.0)))
    oe = (sp[0][own] - t0) / 100.0
    print("   owned entry: median %.2f max %.2f us" % (np.median(oe), oe.max()))
