"""Developer aid (pure CPU, no GPU, no library): what would a walk that may END EARLY read?  For a BASELINE config's scene at its
first-pass state: the pruned level-1 list of every query's cell in order of distance from the cell's centre (map_hash.hip:
k_nl_sort), the walk of measure.hip's nl_walk<.., EARLY> emulated on 20 000 sampled queries - read the first B1 entries, stop when
the fifth distance so far is below r(B1) - |q - centre| - and (second part) how well |q - centre| predicts the queries that do not
settle, and what a larger first batch for them would buy.  profiles/round5/r05p_sorted_lists.txt quotes its output.
Usage: python tools/early_exit_estimate.py [cfg=2]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge; ge.load_package()
from malio_amd import scenes
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
sc = scenes.make_scene(cfg=cfg)
L = sc["L"]; st = scenes.unpack_state(sc["state0"], L)
cf = np.float32(1.125); inv_cf = np.float32(1.0)/cf
scan = sc["scan"]; lid = scan[:, 8].astype(np.int64); pb = scan[:, 0:3].astype(np.float64)
Rw = scenes.q_to_R(st["rot"]); X = np.zeros_like(pb)
for l in range(L):
    m = lid == l
    y = pb[m] @ scenes.q_to_R(st["offR"][l]).T + st["offT"][l][None, :]
    if l > 0:
        tc = sc["temporal_comp"][l-1]; y = y @ scenes.q_to_R(tc[0:4]).T + tc[4:7][None, :]
    X[m] = y
pw = (X @ Rw.T + st["pos"][None, :]).astype(np.float32)
qc = np.floor(pw*inv_cf).astype(np.int64)
B = 1 << 20
def key3(c): return ((c[:,0]+B)&0x1FFFFF) | (((c[:,1]+B)&0x1FFFFF)<<21) | (((c[:,2]+B)&0x1FFFFF)<<42)
qkey = key3(qc); ukeys, qinv = np.unique(qkey, return_inverse=True)
mp = sc["map"][:, 0:3].astype(np.float32); g = mp*inv_cf; mi = np.floor(g).astype(np.int64); f = (g-mi).astype(np.float32)
pts, cells = [], []
for dx in (-1,0,1):
  for dy in (-1,0,1):
    for dz in (-1,0,1):
      a2 = np.zeros(len(mp), np.float32)
      for d, ff in ((dx,f[:,0]),(dy,f[:,1]),(dz,f[:,2])):
        if d > 0: a2 += (1-ff)**2
        elif d < 0: a2 += ff**2
      mem = np.nonzero(a2 <= 1.00002)[0]
      k = key3(mi[mem] + np.array([dx,dy,dz])[None,:])
      pos = np.searchsorted(ukeys, k); pos[pos>=len(ukeys)] = 0
      hit = ukeys[pos] == k
      pts.append(mem[hit]); cells.append(pos[hit])
pts = np.concatenate(pts); cells = np.concatenate(cells)
# center of each cell
uc = np.zeros((len(ukeys),3)); 
first = np.zeros(len(ukeys), np.int64); first[qinv] = np.arange(len(qinv))
cc = (qc[first].astype(np.float64)+0.5)*float(cf)
cd = np.linalg.norm(mp[pts].astype(np.float64) - cc[cells], axis=1)
o = np.lexsort((cd, cells)); pts, cells, cd = pts[o], cells[o], cd[o]
start = np.searchsorted(cells, np.arange(len(ukeys))); end = np.searchsorted(cells, np.arange(len(ukeys)), side='right')
rng = np.random.default_rng(0)
sample = rng.choice(len(pw), 20000, replace=False)
for B1 in (16, 24, 32):
  stop = 0; lines_now = 0; lines_new = 0; n=0; stop_second=0
  for qi in sample:
    c = qinv[qi]; s, e = start[c], end[c]; cnt = e-s
    if cnt < 5: continue
    n += 1
    P = mp[pts[s:e]].astype(np.float64); d = np.linalg.norm(P - pw[qi].astype(np.float64), axis=1)
    dq = np.linalg.norm(pw[qi].astype(np.float64)-cc[c])
    ln_all = -(-(cnt+1)*16//128)
    lines_now += ln_all
    pos = B1-1  # header takes one slot
    done = False
    while pos < cnt:
      d5 = np.sort(d[:pos])[4] if pos >= 5 else np.inf
      if d5 < cd[s+pos-1] - dq - 1e-4: done = True; break
      pos += B1
    rd = min(pos, cnt)
    lines_new += -(-(rd+1)*16//128)
    if done: stop += 1
  print("cfg %d B1=%d: queries %d, stop early %.1f %%, list lines per query now %.2f -> %.2f" % (cfg, B1, n, 100.0*stop/n, lines_now/n, lines_new/n))

# --- predictor analysis for B1 = 32 (31 usable: no header in the built version -> 32)
print("---- predictor (batch of 32 entries, no header)")
rows = []
for qi in sample:
    c = qinv[qi]; s, e = start[c], end[c]; cnt = e - s
    if cnt <= 32: continue
    P = mp[pts[s:e]].astype(np.float64); d = np.linalg.norm(P - pw[qi].astype(np.float64), axis=1)
    dq = np.linalg.norm(pw[qi].astype(np.float64) - cc[c])
    d5 = np.sort(d[:32])[4]
    r = cd[s + 31]
    rows.append((dq, d5, r, cnt, d5 < r - dq - 1e-4))
R = np.array(rows)
print("queries with lists > 32: %d of %d; early exit fails for %.2f %% of them" % (len(R), len(sample), 100 * (1 - R[:, 4].mean())))
fail = R[:, 4] == 0
for T in (0.5, 0.55, 0.6, 0.65, 0.7, 0.75, 0.8):
    pred = R[:, 0] > T
    print("  dq > %.2f: predicted %.1f %% of queries; failures not predicted %.2f %% of all queries" % (T, 100 * pred.mean(), 100 * (fail & ~pred).sum() / len(sample)))
# margin-based predictor: r - dq small
for T in (0.5, 0.6, 0.7, 0.8, 0.9):
    pred = (R[:, 2] - R[:, 0]) < T
    print("  r - dq < %.2f: predicted %.1f %%; failures not predicted %.2f %% of all" % (T, 100 * pred.mean(), 100 * (fail & ~pred).sum() / len(sample)))
print("---- first batch of NB entries (dq > T: NB2 entries)")
def fails(qi, nb):
    c = qinv[qi]; s, e = start[c], end[c]; cnt = e - s
    if cnt <= nb: return False, cnt
    P = mp[pts[s:s+nb]].astype(np.float64); d = np.linalg.norm(P - pw[qi].astype(np.float64), axis=1)
    dq = np.linalg.norm(pw[qi].astype(np.float64) - cc[c])
    d5 = np.sort(d)[4]
    return not (d5 < cd[s + nb - 1] - dq - 1e-4), nb
DQ = np.array([np.linalg.norm(pw[qi].astype(np.float64) - cc[qinv[qi]]) for qi in sample])
for nb1, nb2, T in ((32, 32, 9), (40, 40, 9), (32, 40, 0.6), (32, 40, 0.5), (32, 48, 0.6), (32, 48, 0.65), (24, 40, 0.5), (48, 48, 9)):
    nf = 0; ln = 0
    for k, qi in enumerate(sample):
        nb = nb2 if DQ[k] > T else nb1
        f, rd = fails(qi, nb)
        nf += f; ln += -(-rd * 16 // 128)
    p = nf / len(sample)
    print("  %d / %d (dq > %.2f): failures %.2f %% -> workgroups with a straggler %.0f %%; list lines per query %.2f" % (nb1, nb2, T, 100 * p, 100 * (1 - (1 - p) ** 64), ln / len(sample)))
