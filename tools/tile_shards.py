"""Developer aid (pure CPU): what does a tile partition of BASELINE config 4 cost in memory and balance?  For G = 2 / 4 / 8
shards, tile edges 12 ... 64 m and both tile shapes (cubes, malio_set_partition; columns, malio_set_partition_shape with
MALIO_TILE_COLUMNS): replication = map points stored over all shards / map points (a shard stores its tiles + a 2.3 m halo
of whole voxels, malio_part_stores), the largest shard's share of the map, and the balance of the scan (max / mean points
served, malio_part_owner on the scan's world points). The host functions are the library's own ownership arithmetic.
Usage: python tools/tile_shards.py [cfg=4] [map sample=1000000]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge; ge.load_package()
from malio_amd import capi, scenes
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 4
nsamp = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
sc = scenes.make_scene(cfg=cfg)
L = sc["L"]
m = sc["map"][:, :3]
rng = np.random.default_rng(1)
ms = m[rng.permutation(len(m))[:nsamp]] if len(m) > nsamp else m
# world points of the scan under the prior state (what phase A of the first pass decides ownership from)
st = scenes.unpack_state(sc["state0"], L)
scan = sc["scan"]; lid = scan[:, 8].astype(int); pb = scan[:, :3].astype(np.float64)
X = np.zeros_like(pb)
for l in range(L):
    k = lid == l
    y = pb[k] @ scenes.q_to_R(st["offR"][l]).T + st["offT"][l][None, :]
    if l > 0:
        tc = sc["temporal_comp"][l - 1]
        y = y @ scenes.q_to_R(tc[0:4]).T + tc[4:7][None, :]
    X[k] = y
pw = (X @ scenes.q_to_R(st["rot"]).T + st["pos"][None, :]).astype(np.float32)
fs = float(sc["params"]["filter_size_map"])
ext = m.max(0) - m.min(0)
print("config %d: map %d points (%d sampled), extent %.0f x %.0f x %.0f m; scan %d points" % (cfg, len(m), len(ms), ext[0], ext[1], ext[2], len(pw)))
print("%-8s %3s %6s | %-28s | %-28s" % ("shape", "G", "tile", "replication  largest shard", "scan balance max/mean"))
for columns in (False, True):
    for G in (2, 4, 8):
        for tile in (12.0, 16.0, 24.0, 32.0, 48.0, 64.0):
            stores = np.stack([capi.part_stores(ms, r, G, tile, fs, columns) for r in range(G)])
            owner = capi.part_owner(pw, G, tile, columns)
            served = np.bincount(owner, minlength=G)
            print("%-8s %3d %6.0f | %5.2f x       %5.1f %% of the map | %.2f" % (
                "columns" if columns else "cubes", G, tile, stores.sum() / len(ms), 100.0 * stores.sum(1).max() / len(ms), served.max() / served.mean()))
