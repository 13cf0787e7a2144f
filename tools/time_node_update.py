"""Developer aid: malio_node_update_iterated on G shards that share this one GPU (kernels of different shards serialise, so
this is the host side of the node update more than its GPU side) against one engine.  CFG (default 2), G (default 3)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge; ge.load_package()
from malio_amd import capi, scenes
cfg, G = int(os.environ.get("CFG", "2")), int(os.environ.get("G", "3"))
sc = scenes.make_scene(cfg=cfg)
one = capi.Engine(sc["params"]); one.map_build(sc["map"])
res = {}
for part, gated in ((capi.PART_SCAN, 1), (capi.PART_SCAN, 0), (capi.PART_TILES, 1), (capi.PART_TILES, 0)):
    nd = capi.Node(sc["params"], [0] * G, partition=part, tile_m=16.0)
    if gated and G > 3:
        nd.close()
        continue  # (more than three shards on one device share hardware queues: the node updates pass by pass, host/node.cpp)
    nd.set_option("node_gated", gated)  # the gated chain on every shard / one pass at a time
    nd.map_build(sc["map"])
    ts, to = [], []
    for k in range(14):
        nd.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"]); nd.measure(sc["state0"], True)
        t = time.perf_counter(); v = nd.update_iterated(sc["state0"], sc["P0"]); ts.append(time.perf_counter() - t)
        one.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"]); one.measure(sc["state0"], True)
        t = time.perf_counter(); u = one.update_iterated(sc["state0"], sc["P0"]); to.append(time.perf_counter() - t)
    assert v["passes"] == u["passes"] and np.abs(v["state"] - u["state"]).max() < 1e-8
    print("cfg %d, %d %s shards on one GPU, %s: node update %.1f us (min %.1f), one engine %.1f us, passes %d" % (
        cfg, G, "scan" if part == capi.PART_SCAN else "tile", "gated chain" if gated else "pass by pass", np.median(ts[3:]) * 1e6,
        min(ts) * 1e6, np.median(to[3:]) * 1e6, v["passes"]))
    nd.close()
