"""Developer aid: how uneven is the level-1 list walk between the workgroups of a search pass?  Per query the length of its
cell's list (points of the 3x3x3 block of 1.125 m cells around the query's cell, unpruned: an upper bound), queries grouped
as the scan's grouping does (LiDAR, then cell), 64 per workgroup, 16 per wave, 4 lanes per query, 8 entries per lane and
batch: a wave walks ceil(max length of its 16 queries / 32) batches, a workgroup as many as its slowest wave.   [cfg]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge; ge.load_package()
from malio_amd import capi, scenes
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
sc = scenes.make_scene(cfg=cfg)
e = capi.Engine(sc["params"]); e.map_build(sc["map"]); e.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"])
e.measure(sc["state0"], True)
g = e.scan_get()
w = g["world"].astype(np.float64); lid = sc["scan"][:, 4].astype(np.int64) if sc["scan"].shape[1] > 4 else np.zeros(len(w), np.int64)
cf = 1.125
mc = np.floor(sc["map"][:, :3].astype(np.float64) / cf).astype(np.int64)
off = mc.min(0) - 2; mc -= off
dims = mc.max(0) + 3
key = (mc[:, 0] * dims[1] + mc[:, 1]) * dims[2] + mc[:, 2]
cnt = np.bincount(key, minlength=int(dims.prod()))
qc = np.floor(w / cf).astype(np.int64) - off
ok = ((qc >= 1) & (qc < dims - 1)).all(1)
L = np.zeros(len(w), np.int64)
for dx in (-1, 0, 1):
    for dy in (-1, 0, 1):
        for dz in (-1, 0, 1):
            k = ((qc[ok, 0] + dx) * dims[1] + qc[ok, 1] + dy) * dims[2] + qc[ok, 2] + dz
            L[ok] += cnt[k]
qk = (qc[:, 0] * dims[1] + qc[:, 1]) * dims[2] + qc[:, 2]
order = np.lexsort((qk, lid))
Ls = L[order]
n = len(Ls) // 64 * 64
per_wave = np.ceil(Ls[:n].reshape(-1, 16).max(1) / 32.0)
per_wg = per_wave.reshape(-1, 4).max(1)
print("cfg %d: list length per query (3x3x3 block, unpruned): median %d  p90 %d  p99 %d  max %d" % (cfg, np.median(L), np.percentile(L, 90), np.percentile(L, 99), L.max()))
print("batches per wave: mean %.2f  median %d  p90 %d  p99 %d  max %d" % (per_wave.mean(), np.median(per_wave), np.percentile(per_wave, 90), np.percentile(per_wave, 99), per_wave.max()))
print("batches per workgroup (slowest wave): mean %.2f  median %d  p90 %d  p99 %d  max %d  (%d workgroups)" % (per_wg.mean(), np.median(per_wg), np.percentile(per_wg, 90), np.percentile(per_wg, 99), per_wg.max(), len(per_wg)))
print("if every query walked alone: batches mean %.2f" % np.ceil(Ls / 32.0).mean())
