"""Developer aid (variant built with -DMALIO_PHASE_CLOCK): phase stamps of one mid-grid workgroup of k_pass (search pass) and
the grid-wide spread of workgroup exits.   MALIO_LIB=.../variants/phase.so python tools/phase_pass.py [cfg]"""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
ge.load_package()
from malio_amd import capi, scenes
sc = scenes.make_scene(cfg=int(sys.argv[1]) if len(sys.argv) > 1 else 2)
e = capi.Engine(sc["params"]); e.map_build(sc["map"]); e.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"])
for _ in range(20):
    e.measure(sc["state0"], True)
out = (C.c_longlong * 64)()
assert capi.lib().malio_debug_phase(out) == 0
ph = np.array(out[:]).reshape(4, 16)
n0 = ["enter", "A: transform + sync", "B: level-1 search + sync", "B': level 2 for pending", "neighbours gathered", "plane_cov", "QR",
      "normalise+gates", "trace"]
t0 = ph[0][0]
print("k_pass, mid-grid workgroup, cfg %s (us since its entry; step)" % (sys.argv[1] if len(sys.argv) > 1 else 2), e.fuse_stats())
prev = t0
for j in range(1, len(n0)):
    print("   %-34s %6.2f  %6.2f" % (n0[j], (ph[0][j] - t0) / 100.0, (ph[0][j] - prev) / 100.0)); prev = ph[0][j]
# control wave behind search_wg (KS_SPLIT: hand-off to the helper, one barrier, the plane-dependent rest of the row)
ctl = [(0, "(search_wg returned)"), (6, "flag / plane / residual -> LDS"), (1, "barrier passed"), (2, "row finished"), (3, "rows -> LDS"),
       (4, "16 MFMA"), (5, "tile stores issued")]
for k, name in ctl:
    if ph[2][k] > 0:
        print("   %-34s %6.2f  %6.2f" % (name, (ph[2][k] - t0) / 100.0, (ph[2][k] - prev) / 100.0)); prev = ph[2][k]
hl = ["helper: starts (search over)", "helper: unit_cov + traces", "helper: row factors", "helper: barrier passed", "helper: state stored, extrema published"]
if ph[3][0] > 0:
    prev = ph[3][0]
    for k, name in enumerate(hl):
        print("   %-34s %6.2f  %6.2f" % (name, (ph[3][k] - t0) / 100.0, (ph[3][k] - prev) / 100.0)); prev = ph[3][k]
nb = (sc["N"] + 63) // 64 + 4
sp = (C.c_longlong * (14 * nb))()
assert capi.lib().malio_debug_span(sp, nb) == 0
sp = np.array(sp[:], np.int64).reshape(14, nb)
ok = sp[1] > sp[0]
tz = sp[0][ok].min()
ent, ex = (sp[0][ok] - tz) / 100.0, (sp[1][ok] - tz) / 100.0
print("grid: %d workgroups; entry median %.2f last %.2f; control-wave exit first %.2f median %.2f 90%% %.2f last %.2f us" % (
    ok.sum(), np.median(ent), ent.max(), ex.min(), np.median(ex), np.percentile(ex, 90), ex.max()))
tb = (sp[2][ok] - tz) / 100.0
print("end of level-1 search: median %.2f 90%% %.2f max %.2f us" % (np.median(tb), np.percentile(tb, 90), tb.max()))
# who is late?  exit time of the control wave by XCD (workgroup id mod 8), by LiDAR segment, by position in the grid
b = np.nonzero(ok)[0]
for name, key in (("XCD (id mod 8)", b % 8), ("grid eighth (id * 8 / n)", b * 8 // nb)):
    print("exit by %s: " % name + "  ".join("%d: %.1f/%.1f" % (k, np.median(ex[key == k]), np.percentile(ex[key == k], 95)) for k in sorted(set(key))) + "   (median / p95 us)")
late = b[ex > np.percentile(ex, 97)]
print("the latest 3 %% of the workgroups: ids %s ...; XCD histogram %s" % (late[:16].tolist(), np.bincount(late % 8, minlength=8).tolist()))
print("end of level-1 search by XCD: " + "  ".join("%d: %.1f/%.1f" % (k, np.median(tb[b % 8 == k]), np.percentile(tb[b % 8 == k], 95)) for k in range(8)))
pend = sp[3][ok]
print("pending queries per workgroup (level 2 + level 1 again): mean %.2f" % pend.mean())
print("workgroups with level-2 queries: %d; exit median with / without: %.1f / %.1f us" % ((pend > 0).sum(), np.median(ex[pend > 0]) if (pend > 0).any() else float("nan"), np.median(ex[pend == 0])))

# per wave: end of the directory probe / of the level-1 walk (us since the first workgroup's entry); the workgroup waits for its slowest wave
pr = (sp[8:12][:, ok] - tz) / 100.0
wk = (sp[4:8][:, ok] - tz) / 100.0
en = ent
print("directory probe (from entry): median of wave means %.2f; slowest wave of a workgroup: median %.2f p90 %.2f max %.2f" % (
    np.median(pr.mean(0) - en), np.median(pr.max(0) - en), np.percentile(pr.max(0) - en, 90), (pr.max(0) - en).max()))
d = wk - pr
print("walk after the probe, per wave: median %.2f p90 %.2f max %.2f; slowest wave of a workgroup: median %.2f p90 %.2f max %.2f" % (
    np.median(d), np.percentile(d, 90), d.max(), np.median(d.max(0)), np.percentile(d.max(0), 90), d.max(0).max()))
print("spread inside a workgroup (slowest - fastest wave at the end of the walk): median %.2f p90 %.2f max %.2f" % (
    np.median(wk.max(0) - wk.min(0)), np.percentile(wk.max(0) - wk.min(0), 90), (wk.max(0) - wk.min(0)).max()))
lt = tb > np.percentile(tb, 95)
print("the 5 %% of workgroups whose level-1 search ends last: probe (slowest wave) median %.2f, walk (slowest wave) median %.2f, spread median %.2f; the others: %.2f / %.2f / %.2f" % (
    np.median((pr.max(0) - en)[lt]), np.median(d.max(0)[lt]), np.median((wk.max(0) - wk.min(0))[lt]),
    np.median((pr.max(0) - en)[~lt]), np.median(d.max(0)[~lt]), np.median((wk.max(0) - wk.min(0))[~lt])))
if os.environ.get("PHASE_DUMP"):
    np.save(os.environ["PHASE_DUMP"], sp)
