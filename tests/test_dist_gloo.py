"""The N > 1 path on CPU: world_size-2 gloo processes, each owning half of the scan, the staged
measure (MAX all-reduce of the FIC extrema, SUM all-reduce of the per-LiDAR normal-equation blocks) and
the iterated update through malio_ieskf_step, checked against the single-process oracle on the full scan.
The per-shard compute is an oracle-backed stand-in for the HIP stages (no GPU here); everything around it -
sharding, collectives, assemble(), the update loop - is the product's ma-lio_amd/dist.py."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleShardBackend:
    """stage1/stage2 semantics of csrc/measure.hip on a shard, computed with the CPU oracle."""

    def __init__(self, orc, sc, shard):
        self.o = orc.Oracle(sc["params"], threads=1)
        self.o.map_build(sc["map"])
        self.o.scan_set(sc["scan"][shard], sc["tables"], sc["temporal_comp"])
        self.L, self.params = sc["L"], sc["params"]
        self.lid = sc["scan"][shard][:, 8].astype(int)
        self.n_local = len(shard)

    def stage1(self, state, converge):
        self.state, self.converge = np.array(state), converge
        self.o.set_override(None)
        r = self.o.h_share_model(state, converge)
        mm = self.o.last_minmax()
        return torch.tensor([mm[0], -mm[1], mm[2], -mm[3], float(r["M"])], dtype=torch.float64)

    def stage2(self, mm):
        g = mm.numpy()
        # the pass is idempotent at a fixed state: redo it with the all-reduced extrema, no localization weight
        self.o.set_override([g[0], -g[1], g[2], -g[3]], skip_loc_weight=True)
        r = self.o.h_share_model(self.state, self.converge)
        self.o.set_override(None)
        L = self.L
        sums = np.zeros((L, 97))
        if r["valid"]:
            sel = self.o.scan_get()["selected"].astype(bool)
            lid = self.lid[sel]
            Rc = np.where(r["R"] < 1e-4, 1e-3, r["R"])
            iu = np.triu_indices(12)
            for l in range(L):
                m = lid == l
                cols = list(range(6)) + [6 + 3 * l + k for k in range(3)] + [6 + 3 * (L + l) + k for k in range(3)]
                U = r["h_x"][m][:, cols]
                YtX = (U.T / Rc[m]) @ U
                sums[l, :78] = YtX[iu]
                sums[l, 78:90] = (U.T / Rc[m]) @ r["h"][m]
                N = U[:, :3].T @ U[:, :3]
                sums[l, 90:96] = [N[0, 0], N[1, 1], N[2, 2], N[0, 1], N[0, 2], N[1, 2]]
                sums[l, 96] = m.sum()
        return torch.tensor(sums.reshape(-1), dtype=torch.float64)


def _worker(rank, world, port, L, q):
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    ge.load_package()
    from malio_amd import dist as mdist, scenes
    from oracle import orc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc = scenes.make_scene(seed=301 + L, N=1600, Nmap=30000, L=L)
    shard = np.arange(rank, sc["N"], world)  # interleaved shards: both ranks see every LiDAR
    be = OracleShardBackend(orc, sc, shard)
    one = mdist.sharded_measure(be, sc["state0"], True)
    upd = mdist.sharded_update_iterated(be, sc["state0"], sc["P0"])
    q.put((rank, one, upd))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("L", [1, 3])
def test_sharded_update_matches_single_process(orc, scenes, capi, L):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + L
    procs = [ctx.Process(target=_worker, args=(r, world, port, L, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sc = scenes.make_scene(seed=301 + L, N=1600, Nmap=30000, L=L)
    o = orc.Oracle(sc["params"], threads=2)
    o.map_build(sc["map"])
    o.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"])
    r = o.h_share_model(sc["state0"], True)
    Rc = np.where(r["R"] < 1e-4, 1e-3, r["R"])
    HtH, Hth = (r["h_x"].T / Rc) @ r["h_x"], (r["h_x"].T / Rc) @ r["h"]
    o.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"])
    v = o.update_iterated(sc["state0"], sc["P0"])
    for rank, one, upd in res:
        assert one["valid"] and one["M"] == r["M"]
        assert one["w_loc"] == pytest.approx(r["weight"], rel=1e-10)
        assert np.abs(one["HtRinvH"] - HtH).max() <= 1e-10 * np.abs(HtH).max()
        assert np.abs(one["HtRinvh"] - Hth).max() <= 1e-10 * np.abs(Hth).max()
        assert (upd["passes"], upd["searches"], upd["M"]) == (v["passes"], v["searches"], v["M"])
        assert np.abs(upd["state"] - v["state"]).max() < 1e-8
    # every rank ran the same algebra on the same reduced sums: bit-identical results
    assert np.array_equal(res[0][2]["state"], res[1][2]["state"]) and np.array_equal(res[0][2]["P"], res[1][2]["P"])


def test_assemble_matches_rows(orc, scenes):
    """dist.assemble() on exact per-LiDAR sums == the reference accumulation on full rows."""
    from malio_amd import dist as mdist
    sc = scenes.make_scene(seed=311, N=900, Nmap=20000, L=3)
    be = OracleShardBackend(orc, sc, np.arange(sc["N"]))
    out = mdist.sharded_measure(be, sc["state0"], True)
    o = orc.Oracle(sc["params"], threads=1)
    o.map_build(sc["map"])
    o.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"])
    r = o.h_share_model(sc["state0"], True)
    Rc = np.where(r["R"] < 1e-4, 1e-3, r["R"])
    HtH = (r["h_x"].T / Rc) @ r["h_x"]
    assert out["M"] == r["M"] and np.abs(out["HtRinvH"] - HtH).max() <= 1e-11 * np.abs(HtH).max()
