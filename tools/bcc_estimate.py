"""Developer aid (pure CPU): would a SECOND level-1 lattice, shifted by half a cell in x, y and z, let the ordered-list walk read
fewer lines? A query takes the lattice whose cell centre is nearer (|q - c| <= 0.56 cell edges instead of 0.87), so the early-exit
bound r - |q - c| (measure.hip: nl_walk<.., EARLY>) is tighter for the same number of entries read. For a BASELINE config's scene:
the share of queries that settle after a first batch of 16 / 24 / 32 entries and the list lines read per query, one lattice
against two. The price of the second lattice is 2 x the level-1 list memory and maintenance; nothing here is built.
Usage: python tools/bcc_estimate.py [cfg=2]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge; ge.load_package()
from malio_amd import scenes
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
sc = scenes.make_scene(cfg=cfg)
L = sc["L"]; st = scenes.unpack_state(sc["state0"], L)
cf = 1.125
scan = sc["scan"]; lid = scan[:, 8].astype(np.int64); pb = scan[:, 0:3].astype(np.float64)
Rw = scenes.q_to_R(st["rot"]); X = np.zeros_like(pb)
for l in range(L):
    m = lid == l
    y = pb[m] @ scenes.q_to_R(st["offR"][l]).T + st["offT"][l][None, :]
    if l > 0:
        tc = sc["temporal_comp"][l - 1]; y = y @ scenes.q_to_R(tc[0:4]).T + tc[4:7][None, :]
    X[m] = y
pw = X @ Rw.T + st["pos"][None, :]
mp = sc["map"][:, 0:3].astype(np.float64)
rng = np.random.default_rng(0)
sample = rng.choice(len(pw), 20000, replace=False)
B = 1 << 20
def key3(c): return ((c[:, 0] + B) & 0x1FFFFF) | (((c[:, 1] + B) & 0x1FFFFF) << 21) | (((c[:, 2] + B) & 0x1FFFFF) << 42)

def lattice(off):
    """per sampled query: distance to its cell centre, and its cell's pruned list as distances (from the centre, from the query) in list order"""
    q = pw[sample] - off
    qc = np.floor(q / cf).astype(np.int64)
    ukeys, qinv = np.unique(key3(qc), return_inverse=True)
    g = (mp - off) / cf; mi = np.floor(g).astype(np.int64); f = g - mi
    pts, cells = [], []
    for dx in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dz in (-1, 0, 1):
                a2 = np.zeros(len(mp))
                for d, ff in ((dx, f[:, 0]), (dy, f[:, 1]), (dz, f[:, 2])):
                    if d > 0: a2 += (1 - ff) ** 2
                    elif d < 0: a2 += ff ** 2
                mem = np.nonzero(a2 <= 1.00002)[0]  # nl_member: within one cell edge of the cell
                k = key3(mi[mem] + np.array([dx, dy, dz])[None, :])
                pos = np.searchsorted(ukeys, k); pos[pos >= len(ukeys)] = 0
                hit = ukeys[pos] == k
                pts.append(mem[hit]); cells.append(pos[hit])
    pts = np.concatenate(pts); cells = np.concatenate(cells)
    first = np.zeros(len(ukeys), np.int64); first[qinv] = np.arange(len(qinv))
    cc = (qc[first] + 0.5) * cf + off
    cd = np.linalg.norm(mp[pts] - cc[cells], axis=1)
    o = np.lexsort((cd, cells)); pts, cells, cd = pts[o], cells[o], cd[o]
    start = np.searchsorted(cells, np.arange(len(ukeys))); end = np.searchsorted(cells, np.arange(len(ukeys)), side="right")
    dq = np.linalg.norm(pw[sample] - cc[qinv], axis=1)
    return dict(dq=dq, qinv=qinv, start=start, end=end, pts=pts, cd=cd)

A, Bh = lattice(np.zeros(3)), lattice(np.full(3, 0.5 * cf))
use_b = Bh["dq"] < A["dq"]
print("cfg %d: |q - c| / cf: one lattice median %.2f p95 %.2f max %.2f; nearer of two: median %.2f p95 %.2f max %.2f (%.0f %% take the shifted one)" % (
    cfg, np.median(A["dq"]) / cf, np.percentile(A["dq"], 95) / cf, A["dq"].max() / cf,
    np.median(np.minimum(A["dq"], Bh["dq"])) / cf, np.percentile(np.minimum(A["dq"], Bh["dq"]), 95) / cf, np.minimum(A["dq"], Bh["dq"]).max() / cf, 100 * use_b.mean()))

def walk(Lt, k, nb):
    c = Lt["qinv"][k]; s, e = Lt["start"][c], Lt["end"][c]; cnt = e - s
    if cnt < 5: return None
    if cnt <= nb: return True, cnt
    d = np.linalg.norm(mp[Lt["pts"][s:s + nb]] - pw[sample[k]], axis=1)
    d5 = np.sort(d)[4]
    ok = d5 < Lt["cd"][s + nb - 1] * 0.9999 - Lt["dq"][k] * 1.0001 - 1e-6
    return ok, (nb if ok else cnt)

for nb in (16, 24, 32):
    for name, pick in (("one lattice ", lambda k: A), ("two lattices", lambda k: Bh if use_b[k] else A)):
        n = st_ = ln = 0
        for k in range(len(sample)):
            r = walk(pick(k), k, nb)
            if r is None: continue
            n += 1; st_ += r[0]; ln += -(-r[1] * 16 // 128)
        p = 1 - st_ / n
        print("  first batch %2d entries, %s: settle %.1f %%; a wave of 16 queries has an unsettled one %.0f %%; list lines per query %.2f" % (
            nb, name, 100 * st_ / n, 100 * (1 - (1 - p) ** 16), ln / n))
