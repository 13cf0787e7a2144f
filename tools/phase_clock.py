"""Developer aid: per-phase time of one mid-grid workgroup of k_plane / k_rows_reduce (needs `make PHASE=1`)."""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import __graft_entry__ as ge
ge.load_package()
from malio_amd import capi, scenes
sc = scenes.make_scene(cfg=int(sys.argv[1]) if len(sys.argv) > 1 else 2)
e = capi.Engine(sc["params"]); e.set_option("fuse", 0); e.map_build(sc["map"]); e.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"])
for _ in range(20):
    e.measure(sc["state0"], True)
e.measure(sc["state0"], True)
out = (C.c_longlong * 64)()
assert capi.lib().malio_debug_phase(out) == 0
ph = np.array(out[:]).reshape(4, 16)
names = {0: ["enter", "A: transform + sync", "B: level-1 search + sync", "C: level-2 for pending", "neighbours gathered", "plane_cov", "QR", "normalise+gates", "trace", "extrema"],
         1: ["enter", "extrema fold", "point_row", "LDS stage + sync", "97 sums", "write partials"]}
for k, nm in names.items():
    t = ph[k][:len(nm)]
    print("kernel", "k_search" if k == 0 else "k_rows_reduce", "(100 MHz ticks -> us)")
    for j in range(1, len(nm)):
        print("   %-40s %6.2f us" % (nm[j], (t[j] - t[j - 1]) / 100.0))
    print("   %-40s %6.2f us" % ("total", (t[len(nm) - 1] - t[0]) / 100.0))

nb = (sc["N"] + 63) // 64
sp = (C.c_longlong * (4 * nb))()
assert capi.lib().malio_debug_span(sp, nb) == 0
sp = np.array(sp[:], np.int64).reshape(4, nb)
t0 = sp[0].min()
ent, ex = (sp[0] - t0) / 100.0, (sp[1] - t0) / 100.0
print("k_search grid: %d workgroups; entry: median %.2f, 90%% %.2f, last %.2f us; exit: first %.2f, median %.2f, 90%% %.2f, last %.2f us; "
      "residence: median %.2f, max %.2f us" % (nb, np.median(ent), np.percentile(ent, 90), ent.max(), ex.min(), np.median(ex),
                                              np.percentile(ex, 90), ex.max(), np.median(ex - ent), (ex - ent).max()))
late = np.argsort(ent)[-8:]
print("   last to enter:", [(int(b), round(float(ent[b]), 2), round(float(ex[b]), 2)) for b in late])
slow = np.argsort(ex)[-8:]
print("   last to exit :", [(int(b), round(float(ent[b]), 2), round(float(ex[b]), 2)) for b in slow])

tb, npend = (sp[2] - t0) / 100.0, sp[3]
print("   end of level-1 search: median %.2f, 90%% %.2f, max %.2f us" % (np.median(tb), np.percentile(tb, 90), tb.max()))
for lo, hi in ((0, 0), (1, 2), (3, 4), (5, 8), (9, 64)):
    m = (npend >= lo) & (npend <= hi)
    if m.any():
        print("   pending level-2 queries %2d..%2d: %4d workgroups, exit median %.2f max %.2f us, after-search part median %.2f us" % (
            lo, hi, int(m.sum()), np.median(ex[m]), ex[m].max(), np.median(ex[m] - tb[m])))
