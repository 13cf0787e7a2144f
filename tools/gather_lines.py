"""CPU estimate behind the map array's cell order (round 6): how many distinct 128-byte lines of the map array (8 float4 slots
each) does the gather of the five neighbours ask for - per query, and per workgroup of 64 queries (what a CU's L1 sees) - when
the map array is in the order the scene generator leaves it (raster), shuffled, or sorted by the Morton code of the level-1
cell / half-cell / quarter-cell?   python tools/gather_lines.py [cfg=2]"""
import os, sys
import numpy as np
from scipy.spatial import cKDTree
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge; ge.load_package()
from malio_amd import scenes

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
sc = scenes.make_scene(cfg=cfg)
L = sc["L"]
st = scenes.unpack_state(sc["state_gt"], L)
R = scenes.q_to_R(st["rot"])
# world points of the scan at the ground truth (the neighbours are what they are to within the prior's centimetres)
pw = np.zeros((sc["N"], 3))
lid = sc["scan"][:, 8].astype(int)
for l in range(L):
    m = lid == l
    p = sc["scan"][m, 0:3].astype(np.float64)
    x = p @ scenes.q_to_R(st["offR"][l]).T + st["offT"][l]
    if l > 0:
        tc = sc["temporal_comp"][l - 1]
        x = x @ scenes.q_to_R(tc[0:4]).T + tc[4:7]
    pw[m] = x @ R.T + st["pos"]
mp = sc["map"][:, :3].astype(np.float64)
_, nn = cKDTree(mp).query(pw, k=5)
cf = 1.125
cell = np.floor(pw / cf).astype(np.int64)
order = np.lexsort((cell[:, 2], cell[:, 1], cell[:, 0], lid))  # (LiDAR, cell): what the grouping of the scan aims at


def spread(v):
    v = v & 0x3FF
    out = np.zeros_like(v)
    for b in range(10):
        out |= ((v >> b) & 1) << (3 * b)
    return out


def morton_perm(h):
    c = np.floor(mp / h).astype(np.int64)
    key = spread(c[:, 0]) | (spread(c[:, 1]) << 1) | (spread(c[:, 2]) << 2)
    return np.argsort(key, kind="stable")


def report(name, slot_of):
    lines = slot_of[nn] // 8
    per_q = np.mean([len(set(r)) for r in lines[:: max(1, sc["N"] // 20000)]])
    wg = lines[order]
    nwg = sc["N"] // 64
    per_wg = np.mean([len(np.unique(wg[k * 64:(k + 1) * 64])) for k in range(nwg)])
    print("%-28s lines per query %.2f   distinct lines per 64-query workgroup %.1f (%.2f per query)" % (name, per_q, per_wg, per_wg / 64))


def raster_perm(h, major):
    c = np.floor(mp / h).astype(np.int64)
    c -= c.min(0)
    a, b = (1, 0) if major == "y" else (0, 1)
    key = (c[:, a] * 8192 + c[:, b]) * 1024 + c[:, 2]
    return np.argsort(key, kind="stable")


n = mp.shape[0]
report("raster (scene generator)", np.arange(n))
report("shuffled", np.random.default_rng(1).permutation(n))
for h, nm in ((cf, "Morton, level-1 cell"), (cf / 2, "Morton, half-cell"), (cf / 4, "Morton, quarter-cell")):
    perm = morton_perm(h)
    slot_of = np.empty(n, np.int64)
    slot_of[perm] = np.arange(n)
    report(nm, slot_of)
for h, mj, nm in ((cf, "y", "columns (cy, cx, cz), cell"), (cf, "x", "columns (cx, cy, cz), cell"), (cf / 2, "y", "columns (cy, cx, cz), half-cell"), (2 * cf, "y", "columns (cy, cx, cz), 2.25 m")):
    perm = raster_perm(h, mj)
    slot_of = np.empty(n, np.int64)
    slot_of[perm] = np.arange(n)
    report(nm, slot_of)
