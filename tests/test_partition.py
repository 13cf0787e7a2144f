"""Spatial sharding of the map (BASELINE config 4): tile ownership + halo.

CPU: the geometry alone - every map point the reference could accept as a neighbour of a scan point (d2 <= 5,
laserMapping.cpp:587) is stored by the shard that serves that scan point, checked against an exhaustive k-d tree and
against the reference's own ikd-Tree. GPU (marked): shards on one GPU behind a node handle give, point for point, the
bits one engine with the whole map gives, and the same normal equations up to the order of the final additions."""
import numpy as np
import pytest

from conftest import assert_P_close


def _part(capi, name):
    return {"scan": capi.PART_SCAN, "tiles": capi.PART_TILES, "columns": capi.PART_COLUMNS}[name]


@pytest.mark.parametrize("columns", [False, True], ids=["cubes", "columns"])
@pytest.mark.parametrize("world,tile", [(2, 0.0), (3, 12.0), (8, 16.0), (8, 32.0)])
def test_every_acceptable_neighbour_is_shard_local(capi, orc, scenes, world, tile, columns):
    from scipy.spatial import cKDTree
    sc = scenes.make_scene(seed=301, N=6000, Nmap=120000, L=3)
    m = sc["map"][:, :3].astype(np.float32)
    rng = np.random.default_rng(5)
    # world points of a scan: map points jittered by up to 1 m (some outside every wall), plus points on tile faces
    q = (m[rng.integers(0, len(m), 6000)] + rng.uniform(-1, 1, (6000, 3))).astype(np.float32)
    t = tile if tile > 0 else 16.0
    q[:500, 0] = np.round(q[:500, 0] / t) * t                      # exactly on a tile face
    q[500:1000, 1] = np.nextafter(np.round(q[500:1000, 1] / t) * t, -np.inf).astype(np.float32)  # one ulp below it
    owner = capi.part_owner(q, world, tile, columns)
    assert set(np.unique(owner)) <= set(range(world)) and len(np.unique(owner)) == world
    if columns:  # a column owns every height over its square
        q_up = q.copy()
        q_up[:, 2] += 37.0
        assert np.array_equal(capi.part_owner(q_up, world, tile, True), owner)
    stores = np.stack([capi.part_stores(m, r, world, tile, 0.5, columns) for r in range(world)])  # [world, Nmap]
    assert stores.any(0).all()                                       # every map point lives somewhere
    tree = cKDTree(m.astype(np.float64))
    nb = tree.query_ball_point(q.astype(np.float64), np.sqrt(5.0) + 1e-3)
    for i, lst in enumerate(nb):
        assert stores[owner[i], lst].all(), i
    # whole down-sampling voxels: all points of a voxel are stored by the same shards (Add_Points' keeper rule, f-1)
    vox = np.floor(m / np.float32(0.5)).astype(np.int64)
    _, inv = np.unique(vox, axis=0, return_inverse=True)
    order = np.argsort(inv.ravel(), kind="stable")
    same = inv.ravel()[order][1:] == inv.ravel()[order][:-1]
    a, b = order[1:][same], order[:-1][same]
    assert same.sum() > 100 and np.array_equal(stores[:, a], stores[:, b])
    # replication: a point is stored by at most the 8 shards around a tile corner; on average (16 + 2 x 2.55)^3 / 16^3 = 2.3
    # copies at the default edge when the neighbouring tiles all belong to other shards, 1.6 at 32 m
    # columns: no tile above or below - at most the 4 shards around a column's edge, (16 + 2 x 2.55)^2 / 16^2 = 1.74 copies
    assert stores.sum(0).max() <= min(4 if columns else 8, world)
    if world == 8:
        assert stores.sum() / len(m) < ((2.6 if tile == 16.0 else 1.9) if not columns else (1.8 if tile == 16.0 else 1.45))


def test_shard_local_5nn_equals_reference_tree(capi, orc, scenes):
    """The reference's own ikd-Tree on a shard's part of the map returns, for the shard's own queries, the neighbours it
    returns on the whole map whenever the reference would accept them (d2[4] <= 5)."""
    sc = scenes.make_scene(seed=302, N=3000, Nmap=60000, L=2)
    world, tile = 4, 16.0
    full = orc.Oracle(sc["params"], threads=2, use_ref=True)
    full.map_build(sc["map"])
    rng = np.random.default_rng(6)
    q12 = sc["map"][rng.integers(0, sc["Nmap"], 3000)].copy()
    q12[:, :3] += rng.uniform(-0.6, 0.6, (3000, 3)).astype(np.float32)
    owner = capi.part_owner(q12[:, :3], world, tile)
    pf, d2f, _ = full.knn(q12)
    for r in range(world):
        keep = capi.part_stores(sc["map"][:, :3], r, world, tile, float(sc["params"]["filter_size_map"]))
        shard = orc.Oracle(sc["params"], threads=2, use_ref=True)
        shard.map_build(sc["map"][keep])
        mine = owner == r
        ps, d2s, _ = shard.knn(q12[mine])
        acc = d2f[mine][:, 4] <= 5.0
        assert acc.sum() > 50
        assert np.array_equal(d2s[acc], d2f[mine][acc]) and np.array_equal(ps[acc][:, :, :3], pf[mine][acc][:, :, :3])
        assert (d2s[~acc][:, 4] > 5.0).all()  # and what it rejects stays rejected


# ---------------------------------------------------------------------------------------------------------------------
def _single(capi, sc):
    e = capi.Engine(sc["params"], device=0)
    e.map_build(sc["map"])
    e.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"])
    return e


@pytest.mark.gpu
@pytest.mark.parametrize("partition", ["scan", "tiles", "columns"])
@pytest.mark.parametrize("kw", [dict(seed=311, N=6000, Nmap=150000, L=3), dict(seed=312, N=4000, Nmap=90000, L=2, map_unc=True),
                                dict(seed=313, N=5000, Nmap=120000, L=3, kind="tunnel", det_range=500.0)],
                         ids=lambda k: "s%d" % k["seed"])
@pytest.mark.parametrize("early", [None, 0], ids=["whole", "cut"])
def test_node_handle_equals_single_engine(capi, scenes, partition, kw, early):
    """early = 0 (MALIO_OPT_EARLY_MIN_QUERIES): walks of ordered lists end early on the shards and on the single engine, whatever
    the scan's size - the early exit and the cached probes on tile / column shards, where whole workgroups serve nobody."""
    sc = scenes.make_scene(**kw)
    one = capi.Engine(sc["params"], device=0)
    G = 3
    nd = capi.Node(sc["params"], [0] * G, partition=_part(capi, partition), tile_m=12.0)
    if early is not None:
        one.set_option("early_min_queries", early), nd.set_option("early_min_queries", early)
    one.map_build(sc["map"])
    one.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"])
    nd.map_build(sc["map"])
    if partition != "scan":
        sizes = nd.map_sizes()
        assert max(sizes) < sc["Nmap"] and sum(sizes) >= sc["Nmap"]
    nd.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"])
    s2 = sc["state0"].copy()
    s2[0:3] += [0.012, -0.02, 0.006]
    for state, conv in ((sc["state0"], True), (s2, False), (s2, True), (sc["state0"], False)):
        a, b = one.measure(state, conv), nd.measure(state, conv)
        assert (a["valid"], a["M"]) == (b["valid"], b["M"])
        assert a["unit_cov_minmax"] == b["unit_cov_minmax"] and a["R_minmax"] == b["R_minmax"] and a["w_loc"] == pytest.approx(b["w_loc"], rel=1e-12)
        assert np.abs(a["HtRinvH"] - b["HtRinvH"]).max() <= 1e-12 * np.abs(a["HtRinvH"]).max()
        assert np.abs(a["HtRinvh"] - b["HtRinvh"]).max() <= 1e-12 * np.abs(a["HtRinvh"]).max()
        ga, gb = one.scan_get(), nd.scan_get()
        for k in ("selected", "world", "normvec", "res_last", "nearest_cnt", "nearest", "normal_y"):
            assert np.array_equal(ga[k], gb[k]), k  # point for point, bit for bit
    one.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"])
    nd.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"])
    u, v = one.update_iterated(sc["state0"], sc["P0"]), nd.update_iterated(sc["state0"], sc["P0"])
    assert (u["passes"], u["searches"], u["M"]) == (v["passes"], v["searches"], v["M"])
    # Yardstick (as in test_gpu_parity.test_full_size_configs): the sums of the node differ from the single engine's in
    # their last bits (per-shard partial sums, added in shard order). What that does to the result is a property of the
    # reference's formulas (K_x = P_inv HtH cancels ~11 digits; the tunnel axis is only held by the prior), measured here
    # by the single engine itself on the same points uploaded in another order.
    perm = np.random.default_rng(9).permutation(sc["N"])
    one.scan_set(sc["scan"][perm], sc["tables"], sc["temporal_comp"])
    w = one.update_iterated(sc["state0"], sc["P0"])
    dg = np.sqrt(np.abs(np.diag(u["P"])))
    floor_P = (np.abs(w["P"] - u["P"]) / (np.outer(dg, dg) + 1e-300)).max()
    assert np.abs(u["state"] - v["state"]).max() < max(1e-8, 10 * np.abs(w["state"] - u["state"]).max())  # 1e-8: the state tolerance of test_gpu_parity
    assert_P_close(v["P"], u["P"], rel=max(1e-6, 10 * floor_P))
    hits, misses = nd.exchange_stats()
    # every pass but the first of a scan speculates on the extrema; in the gated chain a miss is followed by a repeat of
    # the pass, which speculates (rightly) too
    assert (4 - 1) + (v["passes"] - 1) <= hits + misses <= (4 - 1) + 2 * (v["passes"] - 1)
    nd.close()


@pytest.mark.gpu
@pytest.mark.parametrize("partition", ["scan", "tiles"])
def test_node_gated_update_same_bits_as_pass_by_pass(capi, orc, scenes, partition):
    """MALIO_OPT_NODE_GATED: every shard runs the gated chain (pass 0 through malio_measure_node, then one speculating
    k_pass per unit, the shards' rows meeting in host memory before the next unit's block is published). Same sums, same
    decisions, same bits as the node's pass-by-pass loop - over three scans in a row (the first has no guess of the extrema,
    the others start with one), with every guess forced wrong (each unit is repeated once), and with the gate of ONE shard
    giving up on a stalled thread (every shard then leaves the chain and the pass-by-pass loop redoes the update)."""
    sc = scenes.make_scene(seed=321, N=6000, Nmap=150000, L=3)
    s2 = sc["state0"].copy()
    s2[0:3] += [0.03, -0.02, 0.01]
    runs = {}
    for mode in ("passes", "gated", "gated_bad_guess", "gated_one_shard_stalls"):
        nd = capi.Node(sc["params"], [0] * 3, partition=capi.PART_TILES if partition == "tiles" else capi.PART_SCAN, tile_m=12.0)
        nd.set_option("node_gated", 0 if mode == "passes" else 1)
        if mode == "gated_bad_guess":
            nd.set_option("debug_fuse_bad_guess", 1)
        if mode == "gated_one_shard_stalls":
            nd.set_option("gate_timeout_ms", 40)
            nd.set_option_rank(1, "debug_gate_stall_ms", 150)
        nd.map_build(sc["map"])
        out = []
        for k, x0 in enumerate((sc["state0"], s2, sc["state0"])):
            nd.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"])
            r = nd.update_iterated(x0, sc["P0"])
            assert r["rc"] == 0, (mode, k, r)
            out.append((r, nd.scan_get()))
            if mode == "gated_one_shard_stalls":
                nd.set_option_rank(1, "debug_gate_stall_ms", 0)  # the first scan only
        runs[mode] = out
        ran = [(nd.get_option_rank(r, "debug_node_gated_runs"), nd.get_option_rank(r, "debug_node_gated_redone")) for r in range(3)]
        assert ran == [{"passes": (0, 0), "gated_one_shard_stalls": (3, 1)}.get(mode, (3, 0))] * 3, (mode, ran)
        st = nd.update_stats()  # the same, summed over the shards, where a caller of the node can see it
        assert (st["gated_runs"], st["gated_redone"]) == (sum(a for a, _ in ran), sum(b for _, b in ran)), (mode, st)
        assert st["gate_timeouts"] == (1 if mode == "gated_one_shard_stalls" else 0), (mode, st)
        nd.close()
    ref = runs["passes"]
    assert ref[0][0]["passes"] >= 3
    # ... and the gated node against the ORACLE directly (reference ikd-Tree + restated h_share_model / esekfom loop), scan by
    # scan: same passes, same accepted points, state to 1e-8, covariance like every other update in the suite
    o = orc.Oracle(sc["params"], threads=8, use_ref=True)
    o.map_build(sc["map"])
    perm = np.random.default_rng(5).permutation(sc["N"])
    for (r, g), x0 in zip(runs["gated"], (sc["state0"], s2, sc["state0"])):
        o.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"])
        v = o.update_iterated(x0, sc["P0"])
        assert (r["passes"], r["searches"], r["M"]) == (v["passes"], v["searches"], v["M"])
        assert np.array_equal(g["selected"], o.scan_get()["selected"])
        # (the yardstick of test_gpu_parity: what the ORACLE's own result moves by when the same points come in another order -
        # the node adds its shards' partial sums in rank order)
        o.scan_set(sc["scan"][perm], sc["tables"], sc["temporal_comp"])
        w = o.update_iterated(x0, sc["P0"])
        dg = np.sqrt(np.abs(np.diag(v["P"])))
        floor_P = (np.abs(w["P"] - v["P"]) / (np.outer(dg, dg) + 1e-300)).max()
        assert np.abs(r["state"] - v["state"]).max() < max(1e-8, 10 * np.abs(w["state"] - v["state"]).max())
        assert_P_close(r["P"], v["P"], rel=max(2e-3, 10 * floor_P))
    for mode in ("gated", "gated_bad_guess", "gated_one_shard_stalls"):
        for (r, g), (r0, g0) in zip(runs[mode], ref):
            assert (r["passes"], r["searches"], r["M"]) == (r0["passes"], r0["searches"], r0["M"]), mode
            assert np.array_equal(r["state"], r0["state"]) and np.array_equal(r["P"], r0["P"]), mode
            for key in ("selected", "world", "normvec", "res_last", "nearest_cnt", "nearest", "normal_y"):
                assert np.array_equal(g[key], g0[key]), (mode, key)


@pytest.mark.gpu
def test_node_shards_that_disagree_on_the_update_loop_fail_together_and_at_once(capi, scenes):
    """Every shard decides from its OWN handle's options whether its update runs the gated chain or pass by pass; the two
    loops meet in different exchanges. A node whose shards disagree (here: MALIO_OPT_NODE_GATED off on shard 1 only) must
    not hang in an exchange until it times out: pass 0 goes through malio_measure_node in both loops, carries the loop's
    mode word, and every shard sees the disagreement in that one exchange. With the option restored the node works again."""
    import time
    sc = scenes.make_scene(seed=322, N=3000, Nmap=60000, L=3)
    nd = capi.Node(sc["params"], [0] * 3, partition=capi.PART_SCAN)
    nd.map_build(sc["map"])
    nd.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"])
    good = nd.update_iterated(sc["state0"], sc["P0"])
    assert good["rc"] == 0
    nd.set_option_rank(1, "node_gated", 0)
    nd.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"])
    t = time.time()
    with pytest.raises(capi.MalioError, match="different update loops"):
        nd.update_iterated(sc["state0"], sc["P0"])
    assert time.time() - t < 5.0  # (an exchange that waits for a shard that never comes gives up after 60 s)
    nd.set_option_rank(1, "node_gated", 1)
    nd.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"])
    again = nd.update_iterated(sc["state0"], sc["P0"])
    assert again["rc"] == 0 and np.array_equal(again["state"], good["state"]) and np.array_equal(again["P"], good["P"])
    nd.close()


@pytest.mark.gpu
def test_node_handle_rccl_exchange_one_gpu(capi, scenes):
    """The RCCL carrier end to end (one rank: the box has one GPU): communicator, device row, ncclAllGather on the handle's
    stream, pinned read-back - results equal the plain engine's bit for bit (one rank adds nothing)."""
    sc = scenes.make_scene(seed=314, N=3000, Nmap=60000, L=3)
    one = _single(capi, sc)
    nd = capi.Node(sc["params"], [0], exchange=capi.XCHG_RCCL)
    nd.map_build(sc["map"])
    nd.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"])
    for conv in (True, False, True):
        a, b = one.measure(sc["state0"], conv), nd.measure(sc["state0"], conv)
        assert a["M"] == b["M"] and np.array_equal(a["HtRinvH"], b["HtRinvH"]) and np.array_equal(a["HtRinvh"], b["HtRinvh"])
    u, v = one.update_iterated(sc["state0"], sc["P0"]), nd.update_iterated(sc["state0"], sc["P0"])
    assert np.array_equal(u["state"], v["state"]) and np.array_equal(u["P"], v["P"])
    nd.close()


@pytest.mark.gpu
def test_tile_shards_map_mutations(capi, scenes):
    """Add_Points / Delete_Point_Boxes on tile shards: every shard is handed the whole call, keeps its part, and the
    next search still equals the single engine's."""
    sc = scenes.make_scene(seed=315, N=4000, Nmap=100000, L=3)
    one = _single(capi, sc)
    nd = capi.Node(sc["params"], [0, 0, 0], partition=capi.PART_TILES, tile_m=12.0)
    nd.map_build(sc["map"])
    nd.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"])
    c = sc["state_gt"][0:3]
    box = np.array([[c[0] - 9, c[1] - 30, c[2] - 5, c[0] + 4, c[1] + 2, c[2] + 5]], np.float32)
    assert one.map_delete_boxes(box) > 100 and sum(nd.map_delete_boxes(box)) >= 100
    rng = np.random.default_rng(3)
    new = sc["map"][rng.integers(0, sc["Nmap"], 5000)].copy()
    new[:, :3] += rng.uniform(-0.2, 0.2, (5000, 3)).astype(np.float32)
    one.map_add(new, True), nd.map_add(new, True)
    one.map_add(new[:300] + np.float32(0.01), False), nd.map_add(new[:300] + np.float32(0.01), False)
    a, b = one.measure(sc["state0"], True), nd.measure(sc["state0"], True)
    assert a["M"] == b["M"] and np.abs(a["HtRinvH"] - b["HtRinvH"]).max() <= 1e-12 * np.abs(a["HtRinvH"]).max()
    ga, gb = one.scan_get(), nd.scan_get()
    for k in ("selected", "normvec", "nearest_cnt"):
        assert np.array_equal(ga[k], gb[k]), k
    assert np.array_equal(ga["nearest"][:, :, :3], gb["nearest"][:, :, :3])
    nd.close()


@pytest.mark.gpu
@pytest.mark.parametrize("shape", ["tiles", "columns"])
def test_config4_tile_sharded_8_shards_equals_single_engine(capi, scenes, shape):
    """BASELINE.json configs[3] at FULL size the way the driver's scaling run executes it: the 8 M-point map cut into 16 m
    hashed tiles over 8 shards (here all on GPU 0, one after the other through the node handle), the 200 k-point scan
    served by the shard that owns each point's tile - against ONE engine that holds the whole map. Flags, planes,
    neighbours and Nearest_Points point for point; the sums to their summation order (per-shard partial sums)."""
    sc = scenes.make_scene(cfg=4)
    one = _single(capi, sc)
    nd = capi.Node(sc["params"], [0] * 8, partition=_part(capi, shape), tile_m=16.0)
    nd.map_build(sc["map"])
    sizes = nd.map_sizes()
    assert max(sizes) < 0.5 * sc["Nmap"] and sum(sizes) >= sc["Nmap"]  # every shard holds its tiles + halo only
    # (cubes of 16 m store 2.6 x the map between them - the halo above and below every tile is most of it -, columns 1.8 x)
    assert sum(sizes) < (2.7 if shape == "tiles" else 1.9) * sc["Nmap"], sizes
    nd.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"])
    s2 = sc["state0"].copy()
    s2[0:3] += [0.012, -0.02, 0.006]
    for state, conv in ((sc["state0"], True), (s2, False), (s2, True)):
        a, b = one.measure(state, conv), nd.measure(state, conv)
        assert (a["valid"], a["M"]) == (b["valid"], b["M"]) and a["M"] > 0.8 * sc["N"]
        assert a["unit_cov_minmax"] == b["unit_cov_minmax"] and a["R_minmax"] == b["R_minmax"]
        assert np.abs(a["HtRinvH"] - b["HtRinvH"]).max() <= 1e-12 * np.abs(a["HtRinvH"]).max()
        assert np.abs(a["HtRinvh"] - b["HtRinvh"]).max() <= 1e-12 * np.abs(a["HtRinvh"]).max()
    ga, gb = one.scan_get(), nd.scan_get()
    for k in ("selected", "world", "normvec", "res_last", "nearest_cnt", "nearest", "normal_y"):
        assert np.array_equal(ga[k], gb[k]), k
    one.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"])
    nd.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"])
    u, v = one.update_iterated(sc["state0"], sc["P0"]), nd.update_iterated(sc["state0"], sc["P0"])
    assert (u["passes"], u["searches"], u["M"]) == (v["passes"], v["searches"], v["M"])
    assert np.abs(u["state"] - v["state"]).max() < 1e-7
    nd.close()


def _gpu_count():
    """malio_device_count: the library's own HIP runtime is asked (a second copy of libamdhip64 loaded through ctypes
    would initialise a second runtime in the process, and the one that comes second finds no device)."""
    try:
        from malio_amd import capi as _c
        return int(_c.lib().malio_device_count())
    except Exception:
        return 0


@pytest.mark.gpu
@pytest.mark.parametrize("partition", ["tiles", "scan"])
def test_node_handle_two_gpus_over_rccl(capi, scenes, partition):
    """The first box with two GPUs runs RCCL with world > 1 under pytest: malio_node_create(n_gpus = 2, XCHG_RCCL) -
    one worker thread and one communicator rank per GPU, ncclAllGather of the [sums | extrema] rows on each handle's
    stream - against one engine on GPU 0. (Skipped on the one-GPU development boxes.)"""
    if _gpu_count() < 2:
        pytest.skip("needs two GPUs")
    sc = scenes.make_scene(seed=316, N=20000, Nmap=300000, L=3)
    one = _single(capi, sc)
    nd = capi.Node(sc["params"], [0, 1], partition=capi.PART_TILES if partition == "tiles" else capi.PART_SCAN,
                   exchange=capi.XCHG_RCCL, tile_m=16.0)
    nd.map_build(sc["map"])
    nd.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"])
    for conv in (True, False, True):
        a, b = one.measure(sc["state0"], conv), nd.measure(sc["state0"], conv)
        assert (a["valid"], a["M"]) == (b["valid"], b["M"])
        assert np.abs(a["HtRinvH"] - b["HtRinvH"]).max() <= 1e-12 * np.abs(a["HtRinvH"]).max()
    ga, gb = one.scan_get(), nd.scan_get()
    for k in ("selected", "normvec", "nearest_cnt", "nearest"):
        assert np.array_equal(ga[k], gb[k]), k
    one.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"])
    nd.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"])
    u, v = one.update_iterated(sc["state0"], sc["P0"]), nd.update_iterated(sc["state0"], sc["P0"])
    assert (u["passes"], u["searches"], u["M"]) == (v["passes"], v["searches"], v["M"])
    assert np.abs(u["state"] - v["state"]).max() < 1e-8
    nd.close()


@pytest.mark.gpu
def test_node_update_small_M_on_a_later_pass_leaves_inputs_untouched(capi, scenes):
    """malio_node_update_iterated has no rows path: a pass that accepts fewer points than there are states returns
    MALIO_SMALL_M_FALLBACK - also when it is a LATER pass of the loop, after valid iterations have produced projected
    covariances. The caller redoes the update from (x, P): both must come back exactly as they went in. A pass hook
    empties the map around the scan before the second search pass, so that pass keeps a handful of points."""
    sc = scenes.make_scene(seed=318, N=3000, Nmap=60000, L=3)
    nd = capi.Node(sc["params"], [0, 0], partition=capi.PART_SCAN)
    nd.map_build(sc["map"])
    nd.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"])
    c = sc["state_gt"][0:3]
    # everything but a small cube around the sensor (six boxes = its complement), sized so that between 3 and 34 scan
    # points can keep five neighbours inside sqrt(5) m: fewer accepted points than the 35 states, but not none
    probe = _single(capi, sc)
    probe.measure(sc["state0"], True)
    pg = probe.scan_get()
    sel = pg["selected"] > 0
    far = np.argsort(-np.abs(pg["world"] - c.astype(np.float32)).max(1))  # sparse regions first
    pick = None
    for j in far[sel[far]][:1500]:
        d = np.abs(pg["world"] - pg["world"][j]).max(1)
        for hh in (3.0, 4.0, 5.0, 6.5, 8.0):
            if (sel & (d < hh - 2.3)).sum() >= 3 and (sel & (d < hh + 2.3)).sum() < 35:
                pick = (pg["world"][j].astype(np.float64), hh)
                break
        if pick:
            break
    assert pick is not None
    c, h = pick
    B = 2000.0
    lo, hi = c - B, c + B
    boxes = np.array([[lo[0], lo[1], lo[2], c[0] - h, hi[1], hi[2]], [c[0] + h, lo[1], lo[2], hi[0], hi[1], hi[2]],
                      [c[0] - h, lo[1], lo[2], c[0] + h, c[1] - h, hi[2]], [c[0] - h, c[1] + h, lo[2], c[0] + h, hi[1], hi[2]],
                      [c[0] - h, c[1] - h, lo[2], c[0] + h, c[1] + h, c[2] - h], [c[0] - h, c[1] - h, c[2] + h, c[0] + h, c[1] + h, hi[2]]],
                     np.float32)
    assert (np.abs(sc["map"][:, :3] - c) < h).all(1).sum() >= 5
    seen = []

    def hook(k):
        seen.append(k)
        if k == 2:
            nd.map_delete_boxes(boxes)
    nd.set_pass_hook(hook)
    P0 = sc["P0"].copy()
    v = nd.update_iterated(sc["state0"], P0)
    nd.set_pass_hook(None)
    assert seen[:3] == [0, 1, 2]
    assert v["rc"] == capi.SMALL_M_FALLBACK, v
    assert np.array_equal(v["P"], sc["P0"]) and np.array_equal(v["state"], sc["state0"])
    nd.close()


@pytest.mark.gpu
@pytest.mark.parametrize("partition", ["scan", "tiles"])
def test_node_map_incremental_equals_single_engine(capi, scenes, partition):
    """map_incremental() on the node (laserMapping.cpp:398-446): every GPU classifies the points it serves from its own
    neighbour cache, the lists are merged in scan order and handed to every GPU. Against ONE engine over three consecutive
    scans of a moving sensor: same |PointToAdd| / |PointNoNeedDownsample|, every GPU's map = the single engine's map
    (a replica: all of it; a tile shard: what malio_part_stores keeps), and the next scan's passes agree."""
    G, tile = 3, 12.0
    sc = scenes.make_scene(seed=321, N=6000, Nmap=120000, L=3)
    one = _single(capi, sc)
    nd = capi.Node(sc["params"], [0] * G, partition=capi.PART_TILES if partition == "tiles" else capi.PART_SCAN, tile_m=tile)
    nd.map_build(sc["map"])
    key = lambda p: p[np.lexsort(p[:, [5, 2, 1, 0]].T)][:, [0, 1, 2, 5]]
    state = sc["state0"]
    for turn in range(3):
        s2 = scenes.make_scene(seed=321, N=6000, Nmap=120000, L=3, scan_seed=700 + turn)
        wny = np.random.default_rng(turn).uniform(0, 0.002, s2["N"]).astype(np.float32)
        one.scan_set(s2["scan"], sc["tables"], sc["temporal_comp"])
        nd.scan_set(s2["scan"], sc["tables"], sc["temporal_comp"])
        u, v = one.update_iterated(state, sc["P0"]), nd.update_iterated(state, sc["P0"])
        assert (u["passes"], u["M"]) == (v["passes"], v["M"])
        # the same posterior for both (the node's differs in the last bits: per-shard partial sums)
        a = one.map_incremental(u["state"], True, wny)
        b = nd.map_incremental(u["state"], True, wny)
        assert (a[0], a[1]) == (b[0], b[1]) and a[0] > 20
        if partition == "scan":
            assert a[2] == b[2]
        full = one.map_get()
        for r in range(G):
            mine = nd.map_get(r)
            want = full if partition == "scan" else full[capi.part_stores(full[:, :3], r, G, tile, sc["params"]["filter_size_map"])]
            assert np.array_equal(key(mine), key(want)), (turn, r, mine.shape, want.shape)
        state = u["state"].copy()
        state[0:3] += [0.05, 0.02, 0.0]  # (the next scan's prior)
    nd.close()


@pytest.mark.gpu
@pytest.mark.parametrize("partition", ["scan", "tiles"])
def test_node_resident_front_end_and_nearest_search(capi, scenes, partition):
    """The resident front end on the node (malio_node_undistort_resident / malio_node_scan_set_resident: LiDAR l on GPU
    l % n, filtered clouds concatenated in LiDAR order) against ONE engine's resident front end: the same
    feats_down_body byte for byte, the same entry points, the same update (sums to summation order); and
    malio_node_nearest_search against malio_nearest_search: identical points, distances and counts."""
    from test_frontend_resident import _traj
    G, L, tile = 2, 3, 12.0
    sc = scenes.make_scene(seed=77, N=9000, Nmap=100000, L=L)
    rng = np.random.default_rng(9)
    t0 = 1671631987.6
    traj = _traj(scenes, t0)
    kt, kT = capi.spline_feed(traj)
    beg, end = t0 + 0.05, t0 + 0.15
    _, q_end, p_end = capi.spline_get_pose(kt, kT, end)
    imu_t = traj[::2, 0].copy()
    cp = int(np.searchsorted(imu_t, end, side="right"))
    st = scenes.unpack_state(sc["state_gt"], L)
    leaf = 0.4
    raws = []
    for l in range(L):
        base = sc["scan"][sc["scan"][:, 8] == l]
        raw = np.repeat(base, 3, axis=0).copy()
        raw[:, :3] += rng.normal(0, 0.05, size=(raw.shape[0], 3)).astype(np.float32)
        raw[:, 9] = np.sort(rng.uniform(0, (end - beg) * 1000.0, raw.shape[0])).astype(np.float32)
        raw[:, 5] = rng.uniform(0, 0.01, raw.shape[0]).astype(np.float32)
        raw[:, 8] = rng.uniform(0, 200, raw.shape[0]).astype(np.float32)
        raws.append(raw)
    one = _single(capi, sc)
    nd = capi.Node(sc["params"], [0] * G, partition=capi.PART_TILES if partition == "tiles" else capi.PART_SCAN, tile_m=tile)
    nd.map_build(sc["map"])
    # nearest search: queries all over the map, some far outside it
    q = sc["scan"][:3000].copy()
    q[:, :3] = sc["map"][rng.integers(0, sc["map"].shape[0], 3000), :3] + rng.normal(0, 0.4, (3000, 3)).astype(np.float32)
    q[:50, :3] += 500.0
    p1, d1, c1 = one.nearest_search(q, 5)
    p2, d2, c2 = nd.nearest_search(q, 5)
    np.testing.assert_array_equal(c1, c2)
    np.testing.assert_array_equal(d1, d2)
    np.testing.assert_array_equal(p1[:, :, [0, 1, 2, 5]], p2[:, :, [0, 1, 2, 5]])
    assert (c1 == 5).sum() > 2000 and (c1[:50] == 0).all()
    # front end
    for l in range(L):
        e1, _ = one.undistort_resident(l, raws[l], beg, kt, kT, st["offR"][l], st["offT"][l], q_end, p_end, imu_t, cp)
        e2 = nd.undistort_resident(l, raws[l], beg, kt, kT, st["offR"][l], st["offT"][l], q_end, p_end, imu_t, cp)
        np.testing.assert_array_equal(e1, e2)
    b1 = one.scan_set_resident(leaf, sc["tables"], sc["temporal_comp"])
    b2 = nd.scan_set_resident(leaf, sc["tables"], sc["temporal_comp"])
    np.testing.assert_array_equal(b1, b2)
    assert b1.shape[0] > 4000 and set(np.unique(b1[:, 8]).astype(int)) == {0, 1, 2}
    u, v = one.update_iterated(sc["state0"], sc["P0"]), nd.update_iterated(sc["state0"], sc["P0"])
    assert (u["passes"], u["M"]) == (v["passes"], v["M"]) and u["M"] > 0.3 * b1.shape[0]
    assert np.abs(u["state"] - v["state"]).max() < 1e-9
    np.testing.assert_array_equal(one.scan_get()["selected"], nd.scan_get()["selected"])
    with pytest.raises(RuntimeError):  # the resident clouds were consumed
        nd.scan_set_resident(leaf, sc["tables"], sc["temporal_comp"])
    nd.close()


@pytest.mark.gpu
def test_tile_shard_forgets_cached_probes_of_an_earlier_scan(capi, scenes):
    """MALIO_OPT_PROBE_CACHE on a tile shard (round-5 advisor finding). A workgroup whose 64 points all belong to other shards
    leaves the search at once and probes nothing - so nothing overwrote its points' cached probes, and entries of an EARLIER scan,
    against another map, passed for this scan's as soon as the points crossed into an owned tile of the same cell key between two
    search passes of one update. Staged here: scan A on map A fills the cache; the map is rebuilt (every list moves); the same
    points, installed again, first land a whole number of tiles AND of 8-cell patches away (72 m: same grouping order, tiles of the
    other shard - every workgroup leaves early), then back where scan A had them: same cell keys as the stale entries. The second
    search pass must find what an unpartitioned engine on map B finds."""
    sc = scenes.make_scene(seed=315, N=20000, Nmap=150000, L=3)
    world, tile = 2, 24.0
    one = capi.Engine(sc["params"])
    one.map_build(sc["map"])
    one.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"])
    one.measure(sc["state0"], True)
    w = one.scan_get()["world"]
    t = np.floor(w / tile).astype(np.int64)
    inner = (np.abs(w / tile - np.round(w / tile)) > 1.0 / tile).all(1)  # a metre away from every tile face
    keys, cnt = np.unique(t[inner], axis=0, return_counts=True)
    pick = inner & (t == keys[np.argmax(cnt)]).all(1)
    assert pick.sum() >= 150, pick.sum()
    scan = sc["scan"][pick]
    r0 = int(capi.part_owner(w[pick][:1], world, tile)[0])
    assert (capi.part_owner(w[pick], world, tile) == r0).all()
    shift = None
    for k in (1, -1, 2, -2, 3, -3, 4, -4):
        for ax in (0, 1):
            d = np.zeros(3)
            d[ax] = 72.0 * k  # (3 tiles = 64 level-1 cells)
            if (capi.part_owner((w[pick] + d).astype(np.float32), world, tile) != r0).all():
                shift = d
                break
        if shift is not None:
            break
    assert shift is not None
    away = sc["state0"].copy()
    away[0:3] += shift
    mapB = sc["map"][np.random.default_rng(4).random(sc["Nmap"]) < 0.7]  # (every list starts somewhere else)
    e = capi.Engine(sc["params"])
    e.set_partition(r0, world, tile)
    e.set_option("early_min_queries", 0)
    e.map_build(sc["map"])
    e.scan_set(scan, sc["tables"], sc["temporal_comp"])
    assert e.measure(sc["state0"], True)["M"] > 100  # scan A: every probe cached
    e.map_build(mapB)
    e.scan_set(scan, sc["tables"], sc["temporal_comp"])
    assert e.measure(away, True)["M"] == 0 and not e.scan_owned().any()  # nobody's points here: every workgroup leaves early
    got = e.measure(sc["state0"], True)
    ref = capi.Engine(sc["params"])
    ref.map_build(mapB)
    ref.scan_set(scan, sc["tables"], sc["temporal_comp"])
    want = ref.measure(sc["state0"], True)
    assert e.scan_owned().all() and got["M"] == want["M"] > 100
    a, b = e.scan_get(), ref.scan_get()
    for f in ("nearest_cnt", "nearest", "selected", "normvec", "res_last"):
        assert np.array_equal(a[f], b[f]), f
    assert np.abs(got["HtRinvH"] - want["HtRinvH"]).max() <= 1e-12 * np.abs(want["HtRinvH"]).max()  # (a shard groups its scan differently: summation order)
