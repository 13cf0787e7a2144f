"""Developer aid (pure CPU, no GPU, no library): how much of a search pass' level-1 list traffic do the 64 queries of a
workgroup SHARE?  Emulates the scan grouping of k_sort_count / k_sort_place (LiDAR slot, bucket = column of cells, then
(cell, index)) on a BASELINE config's scene at its first-pass state, forms the pruned level-1 lists' lengths (nl_member:
map points within one cell edge of the cell) and reports, per 64-query workgroup:
  distinct level-1 cells, runs of equal cells in scan order (what an adjacent-compare dedupe sees),
  unique list entries (each distinct list once) vs the per-query sum, the same padded to 16 entries per list
  (what an LDS stage of the lists holds), and how many workgroups exceed a stage of CAP entries.
Usage: python tools/cell_share.py [cfg=2] [cap=800]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge; ge.load_package()
from malio_amd import scenes

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
CAP = int(sys.argv[2]) if len(sys.argv) > 2 else 800
sc = scenes.make_scene(cfg=cfg)
L = sc["L"]
st = scenes.unpack_state(sc["state0"], L)
cf = np.float32(1.125)
inv_cf = np.float32(1.0) / cf

# world points under the first-pass state (double, like k_sort_count)
scan = sc["scan"]
lid = scan[:, 8].astype(np.int64)
pb = scan[:, 0:3].astype(np.float64)
Rw = scenes.q_to_R(st["rot"])
X = np.zeros_like(pb)
for l in range(L):
    m = lid == l
    Rl = scenes.q_to_R(st["offR"][l])
    y = pb[m] @ Rl.T + st["offT"][l][None, :]
    if l > 0:
        tc = sc["temporal_comp"][l - 1]
        y = y @ scenes.q_to_R(tc[0:4]).T + tc[4:7][None, :]
    X[m] = y
pw = (X @ Rw.T + st["pos"][None, :]).astype(np.float32)
qc = np.floor(pw * inv_cf).astype(np.int64)

# scan order: (lid, bucket, cell key (10 bits per axis), index)
cx, cy, cz = qc[:, 0] & 1023, qc[:, 1] & 1023, qc[:, 2] & 1023
cell10 = (cz << 20) | (cy << 10) | cx
bkt = lid * 4096 + ((((cy & 31) << 6) | (cx & 63)) << 1) + (cz & 1)
order = np.lexsort((np.arange(len(lid)), cell10, bkt))

# pruned level-1 list length of every distinct query cell
mp = sc["map"][:, 0:3].astype(np.float32)
g = mp * inv_cf
mi = np.floor(g).astype(np.int64)
f = (g - mi.astype(np.float32)).astype(np.float32)
B = 1 << 20
def key3(c):
    return ((c[:, 0] + B) & 0x1FFFFF) | (((c[:, 1] + B) & 0x1FFFFF) << 21) | (((c[:, 2] + B) & 0x1FFFFF) << 42)
qkey = key3(qc)
ukeys, qinv = np.unique(qkey, return_inverse=True)
cnt = np.zeros(len(ukeys), np.int64)
reach = 1.0 + 1e-5
for dx in (-1, 0, 1):
    for dy in (-1, 0, 1):
        for dz in (-1, 0, 1):
            a2 = np.zeros(len(mp), np.float32)
            for d, ff in ((dx, f[:, 0]), (dy, f[:, 1]), (dz, f[:, 2])):
                if d > 0:
                    a2 += (1 - ff) ** 2
                elif d < 0:
                    a2 += ff ** 2
            mem = a2 <= reach * reach
            k = key3(mi[mem] + np.array([dx, dy, dz])[None, :])
            pos = np.searchsorted(ukeys, k)
            pos[pos >= len(ukeys)] = 0
            hit = ukeys[pos] == k
            np.add.at(cnt, pos[hit], 1)
qlen = cnt[qinv]

# workgroups: 64 consecutive sorted queries inside one LiDAR segment
rows = []
lid_s, key_s, len_s = lid[order], qkey[order], qlen[order]
for l in range(L):
    idx = np.nonzero(lid_s == l)[0]
    for s in range(0, len(idx), 64):
        w = idx[s:s + 64]
        k, ln = key_s[w], len_s[w]
        uk, first = np.unique(k, return_index=True)
        runs = 1 + int(np.count_nonzero(k[1:] != k[:-1]))
        run_first = np.concatenate([[True], k[1:] != k[:-1]])
        pad16 = lambda x: (x + 15) // 16 * 16
        rows.append((len(w), len(uk), runs, int(ln[first].sum()), int(ln.sum()), int(pad16(ln[first]).sum()),
                     int(pad16(ln[run_first]).sum()), int(ln.max())))
r = np.array(rows)
def q(x):
    return "mean %.1f  median %d  p90 %d  p99 %d  max %d" % (x.mean(), np.median(x), np.percentile(x, 90), np.percentile(x, 99), x.max())
print("cfg %d: %d queries, %d workgroups, %d distinct level-1 cells in the scan; list length per query: %s" % (
    cfg, len(lid), len(r), len(ukeys), q(qlen)))
print("per workgroup: distinct cells           %s" % q(r[:, 1]))
print("per workgroup: runs of equal cells      %s" % q(r[:, 2]))
print("per workgroup: unique list entries      %s   (x16 B = %.1f KB mean)" % (q(r[:, 3]), r[:, 3].mean() * 16 / 1024))
print("per workgroup: per-query sum of entries %s   (x16 B = %.1f KB mean)" % (q(r[:, 4]), r[:, 4].mean() * 16 / 1024))
print("per workgroup: stage entries (distinct, padded to 16) %s" % q(r[:, 5]))
print("per workgroup: stage entries (runs, padded to 16)     %s" % q(r[:, 6]))
print("workgroups whose run-staged lists exceed %d entries: %d of %d (%.2f %%)" % (CAP, int((r[:, 6] > CAP).sum()), len(r), 100.0 * (r[:, 6] > CAP).mean()))
print("whole pass: unique list bytes %.1f MB, per-query sum %.1f MB, ratio %.2f" % (r[:, 3].sum() * 16 / 1e6, r[:, 4].sum() * 16 / 1e6, r[:, 4].sum() / max(r[:, 3].sum(), 1)))
slots_now = (np.ceil(len_s / 32.0) * 32).sum()
print("candidate slots per query today (2 batches of 32): %.1f; padded to 16: %.1f; entries: %.1f" % (slots_now / len(lid), ((len_s + 15) // 16 * 16).mean(), len_s.mean()))
