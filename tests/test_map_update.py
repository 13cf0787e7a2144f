"""Map maintenance: Add_Points / Delete_Point_Boxes.

CPU part: the flat-list restatement (oracle/orc_map.cpp) against the reference's own ikd-Tree compiled from
source (oracle/_ref). GPU part: the HIP map (malio_map_add / malio_map_delete_boxes / malio_map_get) against both,
as SETS of valid points (the map order is an implementation detail on either side), followed by a k-NN check so
the rebuilt search structure is covered too.
"""
import numpy as np
import pytest


def _pts(rng, n, lo, hi, cov_scale=0.01):
    p = np.zeros((n, 12), np.float32)
    p[:, :3] = rng.uniform(lo, hi, size=(n, 3)).astype(np.float32)
    p[:, 3] = 1.0
    p[:, 5] = (rng.uniform(0.0, 1.0, n) * cov_scale).astype(np.float32)  # normal_y: the stored uncertainty
    p[:, 8] = rng.uniform(0, 255, n).astype(np.float32)
    return p


def _as_set(p12):
    """Sorted [n,4] view (x, y, z, normal_y) - the fields every implementation stores."""
    a = np.ascontiguousarray(p12[:, [0, 1, 2, 5]], np.float32)
    order = np.lexsort((a[:, 3], a[:, 2], a[:, 1], a[:, 0]))
    return a[order]


def _script(seed, n0=4000, nadd=1500, rounds=4, ds=0.5, extent=12.0, faces=True):
    """A mapping-loop-like call sequence: build, then per round add(downsample), add(no downsample), and
    sometimes a box delete - with duplicates, near-centre points and voxel-boundary coordinates mixed in."""
    rng = np.random.default_rng(seed)
    ops = [("build", _pts(rng, n0, -extent, extent))]
    for r in range(rounds):
        a = _pts(rng, nadd, -extent - 2, extent + 2)
        k = nadd // 10
        # exact duplicates of stored/new points, points at voxel centres (near rule), points on voxel faces
        a[:k, :3] = ops[0][1][rng.integers(0, n0, k), :3]
        cen = (np.floor(a[k:2 * k, :3] / ds) * ds + ds / 2).astype(np.float32)
        a[k:2 * k, :3] = cen + rng.uniform(-0.05, 0.05, size=(k, 3)).astype(np.float32)
        if faces:
            a[2 * k:3 * k, 0] = (np.floor(a[2 * k:3 * k, 0] / ds) * ds).astype(np.float32)
        a[3 * k:4 * k] = a[4 * k:5 * k]  # repeated inside the same call
        ops.append(("add", a, True))
        ops.append(("add", _pts(rng, nadd // 5, -extent, extent), False))
        if r % 2 == 1:
            c = rng.uniform(-extent, extent, size=(2, 3)).astype(np.float32)
            boxes = np.concatenate([c - 3.0, c + 3.0], axis=1).astype(np.float32)
            boxes[0, :3] = np.floor(boxes[0, :3] / ds) * ds  # faces on the voxel lattice: the < / <= edges matter
            ops.append(("del", boxes))
    return ops


def _run(target, ops):
    rets = []
    for op in ops:
        if op[0] == "build":
            target.build(op[1])
        elif op[0] == "add":
            rets.append(int(target.add(op[1], op[2])))
        else:
            rets.append(int(target.delete_boxes(op[1])))
        rets.append(int(target.size()))
    return rets


@pytest.mark.parametrize("seed,ds", [(1, 0.5), (2, 0.5), (3, 0.3), (4, 0.25)])
def test_restatement_matches_reference_tree(orc, seed, ds):
    if not orc.have_ref():
        pytest.skip("oracle/_ref not built")
    ops = _script(seed, ds=ds)
    ref, port = orc.RefTree(ds), orc.VoxMap(ds)
    r_ref, r_port = _run(ref, ops), _run(port, ops)
    assert r_ref == r_port
    np.testing.assert_array_equal(_as_set(ref.flatten()), _as_set(port.flatten()))


def test_restatement_keeper_rule_units(orc):
    """Hand-built voxel [0,0.5)^3: the near-centre / closest rules and the 'nothing happens' branch."""
    ds = 0.5

    def P(x, y, z, cov):
        p = np.zeros((1, 12), np.float32)
        p[0, :3] = (x, y, z)
        p[0, 5] = cov
        return p

    # neither near: d^2(stored) = 3*0.04 = 0.12, d^2(new) = 3*0.0225 = 0.0675 -> the new one is closer and replaces
    m = orc.VoxMap(ds)
    m.build(P(0.05, 0.05, 0.05, 0.5))        # voxel centre is (0.25, 0.25, 0.25)
    assert m.add(P(0.40, 0.40, 0.40, 0.1), True) == 1
    assert m.size() == 1 and np.allclose(m.flatten()[0, :3], 0.40)
    # both near (d^2 < ds/8 = 0.0625): lower normal_y wins even when farther
    m = orc.VoxMap(ds)
    m.build(P(0.30, 0.30, 0.30, 0.9))
    assert m.add(P(0.25, 0.25, 0.25, 0.95), True) == 0 and np.allclose(m.flatten()[0, :3], 0.30)
    assert m.add(P(0.20, 0.20, 0.35, 0.1), True) == 1 and np.allclose(m.flatten()[0, :3], (0.20, 0.20, 0.35))
    # two stored points: the box is always rewritten, even when a stored point wins
    m = orc.VoxMap(ds)
    m.build(np.concatenate([P(0.26, 0.25, 0.25, 0.2), P(0.05, 0.4, 0.1, 0.1)]))
    assert m.add(P(0.45, 0.45, 0.45, 0.0), True) == 1
    assert m.size() == 1 and np.allclose(m.flatten()[0, :3], (0.26, 0.25, 0.25))
    # no-downsample branch returns 0 and appends
    assert m.add(P(0.26, 0.25, 0.25, 0.2), False) == 0 and m.size() == 2
    # box delete: min inclusive, max exclusive
    assert m.delete_boxes(np.array([[0.26, 0.0, 0.0, 0.5, 0.5, 0.5]], np.float32)) == 2
    m.build(P(0.26, 0.25, 0.25, 0.2))
    assert m.delete_boxes(np.array([[0.0, 0.0, 0.0, 0.26, 0.5, 0.5]], np.float32)) == 0


# ------------------------------------------------------------------------------------------------ GPU
class _GpuMap:
    def __init__(self, capi, scenes, ds):
        prm = dict(scenes.make_scene(cfg=1)["params"])
        prm["filter_size_map"] = ds
        self.e = capi.Engine(prm)

    def build(self, p):
        self.e.map_build(p)

    def add(self, p, on):
        return self.e.map_add(p, on)

    def delete_boxes(self, b):
        return self.e.map_delete_boxes(b)

    def size(self):
        return self.e.map_size()

    def flatten(self):
        return self.e.map_get()


@pytest.mark.gpu
@pytest.mark.parametrize("seed,ds", [(1, 0.5), (2, 0.5), (3, 0.3), (5, 0.25)])
def test_gpu_map_update_matches_oracle_and_reference(orc, capi, scenes, seed, ds):
    # points exactly ON a voxel face: exact on the GPU for power-of-two lattices (the shipped 0.5 m); for other
    # sizes a point within one ulp of a face may be attributed to the neighbouring voxel
    ops = _script(seed, ds=ds, faces=float(np.log2(ds)).is_integer())
    gpu, port = _GpuMap(capi, scenes, ds), orc.VoxMap(ds)
    r_gpu, r_port = _run(gpu, ops), _run(port, ops)
    assert r_gpu == r_port
    got = _as_set(gpu.flatten())
    np.testing.assert_array_equal(got, _as_set(port.flatten()))
    if orc.have_ref():
        ref = orc.RefTree(ds)
        assert _run(ref, ops) == r_gpu
        np.testing.assert_array_equal(got, _as_set(ref.flatten()))
        # the rebuilt search structure answers like the reference tree after the same history
        rng = np.random.default_rng(seed + 100)
        q = _pts(rng, 2000, -10, 10)
        _, d2_g, cnt_g = gpu.e.nearest_search(q, 5)
        near = np.zeros((q.shape[0], 5, 12), np.float32)
        d2_r = np.zeros((q.shape[0], 5), np.float32)
        cnt_r = np.zeros(q.shape[0], np.int32)
        import ctypes as C
        ref._l.refikd_knn(ref.h, q.ctypes.data_as(C.POINTER(C.c_float)), q.shape[0], 5,
                          near.ctypes.data_as(C.POINTER(C.c_float)), d2_r.ctypes.data_as(C.POINTER(C.c_float)),
                          cnt_r.ctypes.data_as(C.POINTER(C.c_int)), 4)
        # the GPU search is exact inside radius 2*cell (>= sqrt(5)); compare there
        inside = d2_r < 5.0
        np.testing.assert_array_equal(d2_g[inside], d2_r[inside])


@pytest.mark.gpu
def test_gpu_map_update_mapping_loop_scale(orc, capi, scenes):
    """configs[0]-sized map, one scan's worth of additions: the measurement update keeps matching the oracle
    after the map changed under it."""
    sc = scenes.make_scene(cfg=1)
    ds = float(sc["params"]["filter_size_map"])
    eng = capi.Engine(sc["params"])
    eng.map_build(sc["map"])
    port = orc.VoxMap(ds)
    port.build(sc["map"])
    rng = np.random.default_rng(7)
    new = sc["map"][rng.integers(0, sc["map"].shape[0], 20000)].copy()
    new[:, :3] += rng.normal(0, 0.2, size=(new.shape[0], 3)).astype(np.float32)
    new[:, 5] = (rng.uniform(0, 1, new.shape[0]) * 0.01).astype(np.float32)
    assert eng.map_add(new, True) == port.add(new, True)
    assert eng.map_size() == port.size()
    m_gpu, m_port = eng.map_get(), port.flatten()
    np.testing.assert_array_equal(_as_set(m_gpu), _as_set(m_port))
    # same measurement update on the updated map, both sides
    o = orc.Oracle(sc["params"], threads=4, use_ref=True)
    o.map_build(m_port)
    o.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"])
    eng.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"])
    ro = o.h_share_model(sc["state0"], True)
    rg = eng.measure(sc["state0"], True)
    assert rg["M"] == ro["M"] and rg["valid"] == ro["valid"]
    so, sg = o.scan_get(), eng.scan_get()
    np.testing.assert_array_equal(sg["selected"], so["selected"])
    np.testing.assert_array_equal(sg["res_last"][so["selected"] > 0], so["res_last"][so["selected"] > 0])
    with pytest.raises(RuntimeError):
        eng.map_add(new[:10], True)
        eng.scan_get()  # Nearest_Points now refer to a map that no longer exists


# ------------------------------------------------------------------------------ map_incremental (laserMapping.cpp:398-446)
def _mapinc_case(orc, capi, scenes, cfg, map_filter=None, iterate=False, seed=0, flg=True):
    sc = scenes.make_scene(cfg=cfg)
    ds = float(sc["params"]["filter_size_map"])
    mp = sc["map"] if map_filter is None else sc["map"][map_filter(sc["map"])]
    assert mp.shape[0] > 0
    rng = np.random.default_rng(seed)
    o = orc.Oracle(sc["params"], threads=4, use_ref=True)
    o.map_build(mp)
    o.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"])
    eng = capi.Engine(sc["params"])
    eng.map_build(mp)
    eng.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"])
    if iterate:   # the real sequence: iterated update, then the map takes the scan at the posterior
        uo = o.update_iterated(sc["state0"], sc["P0"])
        eng.update_iterated(sc["state0"], sc["P0"])
        post = uo["state"]
    else:         # one search pass, then a posterior that differs from the search-pass state
        o.h_share_model(sc["state0"], True)
        eng.measure(sc["state0"], True)
        post = np.array(sc["state0"], np.float64).copy()
        post[:3] += (0.03, -0.02, 0.01)
    wny = np.where(np.arange(sc["N"]) < sc["N"] // 3, 0.001, 0.0).astype(np.float32)
    wny[rng.integers(0, sc["N"], 50)] = 0.004
    A, B = o.map_incremental(post, flg, wny)
    na, nn, ret = eng.map_incremental(post, flg, wny)
    assert (na, nn) == (A.shape[0], B.shape[0])
    port = orc.VoxMap(ds)
    port.build(mp)
    assert port.add(A, True) == ret
    port.add(B, False)
    assert eng.map_size() == port.size()
    np.testing.assert_array_equal(_as_set(eng.map_get()), _as_set(port.flatten()))
    return dict(na=na, nn=nn, ret=ret, N=sc["N"], size=port.size())


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,iterate", [(1, False), (1, True), (3, True)])
def test_gpu_map_incremental_matches_oracle(orc, capi, scenes, cfg, iterate):
    r = _mapinc_case(orc, capi, scenes, cfg, iterate=iterate)
    assert r["na"] > 0


@pytest.mark.gpu
def test_gpu_map_incremental_frontier_and_far_map(orc, capi, scenes):
    """Scan points with no map point within sqrt(5) m: points_near[0] of the reference is then far away and
    still steers :421-425 (k_far_nearest: shell search, then the whole-map scan)."""
    cx = scenes.SURFACE_SHIFT[0]
    r1 = _mapinc_case(orc, capi, scenes, 1, map_filter=lambda m: m[:, 0] < cx + 4.0)   # half the scene unmapped
    assert r1["nn"] > 0
    _mapinc_case(orc, capi, scenes, 1, map_filter=lambda m: m[:, 0] > cx + 30.0)       # map tens of metres away
    _mapinc_case(orc, capi, scenes, 1, map_filter=lambda m: np.arange(m.shape[0]) < 3)  # fewer than 5 map points


@pytest.mark.gpu
def test_gpu_map_incremental_before_ekf_init(orc, capi, scenes):
    r = _mapinc_case(orc, capi, scenes, 1, flg=False)   # :411 false -> everything goes through PointToAdd
    assert r["nn"] == 0


@pytest.mark.gpu
def test_gpu_in_place_list_updates_equal_a_full_rebuild(orc, capi, scenes):
    """Mapping-loop regime: a large map, a few thousand changes per call. The changes must reach the neighbour lists
    IN PLACE (slack / tail / tombstones, no rebuild), and every search afterwards must answer exactly like an
    engine whose lists were built from scratch on the resulting map - and like the reference tree."""
    sc = scenes.make_scene(cfg=3)
    ds = float(sc["params"]["filter_size_map"])
    eng = capi.Engine(sc["params"])
    eng.map_build(sc["map"])
    port = orc.VoxMap(ds)
    port.build(sc["map"])
    rng = np.random.default_rng(11)
    q = sc["map"][rng.integers(0, sc["Nmap"], 4000)].copy()
    q[:, :3] += rng.normal(0, 0.4, size=(4000, 3)).astype(np.float32)
    before = eng.debug_counters()
    for r in range(4):
        new = sc["map"][rng.integers(0, sc["Nmap"], 3000)].copy()
        new[:, :3] += rng.normal(0, 0.3, size=(3000, 3)).astype(np.float32)
        new[:, 5] = (rng.uniform(0, 1, 3000) * 0.002).astype(np.float32)
        assert eng.map_add(new, True) == port.add(new, True)
        far = new[:300].copy()
        far[:, :3] += np.float32(400.0)                       # cells that do not exist yet: lists from the tail
        assert eng.map_add(far, False) == port.add(far, False)
        c = (sc["map"][rng.integers(0, sc["Nmap"]), :3] + 0).astype(np.float32)
        box = np.concatenate([c - 2.0, c + 2.0]).astype(np.float32)[None]
        assert eng.map_delete_boxes(box) == port.delete_boxes(box)
        assert eng.map_size() == port.size()
        _, d2_inc, cnt_inc = eng.nearest_search(q, 5)
        fresh = capi.Engine(sc["params"])
        fresh.map_build(port.flatten())
        _, d2_new, cnt_new = fresh.nearest_search(q, 5)
        np.testing.assert_array_equal(d2_inc, d2_new)
        np.testing.assert_array_equal(cnt_inc, cnt_new)
        # the level-1 lists a batch appended to are put back in order of distance from their cell's centre (k_nl_sort):
        # a search may end early in every list that says so
        lo = eng.list_order()
        assert lo["broken"] == 0 and lo["ordered"] >= 0.999 * lo["lists"], lo
    after = eng.debug_counters()
    assert after["rebuilds"] == before["rebuilds"], "the changes were meant to fit in place"
    assert after["inplace"] >= before["inplace"] + 12 and after["tombstones"] > 0
    np.testing.assert_array_equal(_as_set(eng.map_get()), _as_set(port.flatten()))
    # the measurement update on the incrementally maintained map == on a freshly built one
    eng.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"])
    fresh.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"])
    a, b = eng.measure(sc["state0"], True), fresh.measure(sc["state0"], True)
    assert a["M"] == b["M"]
    sa, sb = eng.scan_get(), fresh.scan_get()
    np.testing.assert_array_equal(sa["selected"], sb["selected"])
    np.testing.assert_array_equal(sa["res_last"], sb["res_last"])
    np.testing.assert_allclose(a["HtRinvH"], b["HtRinvH"], rtol=1e-12, atol=1e-9 * np.abs(b["HtRinvH"]).max())
    # many deletions: tombstones pile up past a fifth of the map -> compaction + rebuild, still the same answers
    big = np.array([[-1e4, -1e4, -1e4, 1e4, 1e4, sc["map"][:, 2].mean()]], np.float32)
    assert eng.map_delete_boxes(big) == port.delete_boxes(big)
    _, d2_inc, _ = eng.nearest_search(q, 5)
    fresh = capi.Engine(sc["params"])
    fresh.map_build(port.flatten())
    _, d2_new, _ = fresh.nearest_search(q, 5)
    np.testing.assert_array_equal(d2_inc, d2_new)
    assert eng.debug_counters()["rebuilds"] > after["rebuilds"]


@pytest.mark.gpu
def test_gpu_appends_to_lists_that_hold_tombstones_keep_them_in_order(orc, capi, scenes):
    """A batch appends to lists whose earlier entries died (every voxel-filtered batch does: the points it replaces sit in
    the lists it appends to). k_nl_sort merges the new entries into the live ones - which are still in order among
    themselves - and moves the dead behind them (round 6; rounds 5 sorted such lists from scratch). Tails of every length
    up to the prefix's, dead entries in prefix and tail, lists of one, two and four entries per lane."""
    sc = scenes.make_scene(cfg=1)
    eng = capi.Engine(sc["params"])
    eng.set_option("early_min_queries", 0)   # (walks that rely on the order, at this size too)
    rng = np.random.default_rng(5)
    c = sc["map"][rng.integers(0, sc["map"].shape[0]), :3].astype(np.float32)
    dense = np.repeat(sc["map"][:1], 6000, axis=0).copy()
    dense[:, :3] = (c + rng.uniform(-6, 6, size=(6000, 3))).astype(np.float32)   # +3.5 points per m^3: lists of 100 - 250 entries
    m0 = np.concatenate([sc["map"], dense])
    eng.map_build(m0)
    port = orc.VoxMap(float(sc["params"]["filter_size_map"]))
    port.build(m0)
    q = np.concatenate([dense[:1500], sc["map"][:1500]]).astype(np.float32).copy()
    q[:, :3] += rng.normal(0, 0.2, size=(3000, 3)).astype(np.float32)
    before = eng.debug_counters()
    for r in range(6):
        # holes in dense and thin places, then new points in and around the holes (some of them die in the next round)
        for centre, half in ((c + rng.uniform(-4, 4, 3).astype(np.float32), 0.5 + 0.2 * r),
                             (sc["map"][rng.integers(0, sc["map"].shape[0]), :3].astype(np.float32), 1.5)):
            box = np.concatenate([centre - half, centre + half]).astype(np.float32)[None]
            assert eng.map_delete_boxes(box) == port.delete_boxes(box)
            k = (5, 40, 400, 1500, 12, 90)[r]
            new = np.repeat(sc["map"][:1], k, axis=0).copy()
            new[:, :3] = (centre + rng.uniform(-2 * half, 2 * half, size=(k, 3))).astype(np.float32)
            assert eng.map_add(new, False) == port.add(new, False)
        lo = eng.list_order()
        assert lo["broken"] == 0 and lo["ordered"] >= 0.99 * lo["lists"], lo
        _, d2_inc, cnt_inc = eng.nearest_search(q, 5)
        fresh = capi.Engine(sc["params"])
        fresh.set_option("early_min_queries", 0)
        fresh.map_build(port.flatten())
        _, d2_new, cnt_new = fresh.nearest_search(q, 5)
        np.testing.assert_array_equal(d2_inc, d2_new)
        np.testing.assert_array_equal(cnt_inc, cnt_new)
        eng.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"])
        fresh.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"])
        a, b = eng.measure(sc["state0"], True), fresh.measure(sc["state0"], True)
        assert a["M"] == b["M"]
        np.testing.assert_array_equal(eng.scan_get()["res_last"], fresh.scan_get()["res_last"])
    after = eng.debug_counters()
    assert after["rebuilds"] == before["rebuilds"] and after["tombstones"] > 0, (before, after)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,early", [(1, None), (1, 0), (3, None), (5, None)], ids=["cfg1", "cfg1-cut", "cfg3", "cfg5"])
def test_gpu_ordered_lists_give_the_same_neighbours_as_unordered_ones(capi, scenes, cfg, early):
    """MALIO_OPT_NL_SORTED (default on): the level-1 lists in order of distance from the cell centre let a walk end after
    its first 32 entries when those prove the rest irrelevant. Same five neighbours, same order, same bits as the whole
    walk over unordered lists (rounds 2-4) - through a search pass, an iterated update and a map that changed in place."""
    sc = scenes.make_scene(cfg=cfg)
    res = []
    for srt in (1, 0):
        eng = capi.Engine(sc["params"])
        eng.set_option("nl_sorted", srt)
        if early is not None:  # (config 1's 10 k points are below the default threshold: both engines would walk whole lists)
            eng.set_option("early_min_queries", early)
        eng.map_build(sc["map"])
        eng.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"])
        lo = eng.list_order()
        assert lo["broken"] == 0
        assert (lo["ordered"] >= 0.999 * lo["lists"]) if srt else (lo["ordered"] == 0), lo
        m = eng.measure(sc["state0"], True)
        sg = eng.scan_get()
        u = eng.update_iterated(sc["state0"], sc["P0"])
        na, nn, ret = eng.map_incremental(u["state"], True, np.full(sc["N"], 0.001, np.float32))
        lo2 = eng.list_order()
        assert lo2["broken"] == 0 and ((lo2["ordered"] >= 0.999 * lo2["lists"]) if srt else (lo2["ordered"] == 0)), lo2
        eng.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"])
        m2 = eng.measure(u["state"], True)
        sg2 = eng.scan_get()
        res.append((m, sg, u, (na, nn, ret), m2, sg2, lo["entries"], lo2["entries"]))
    a, b = res
    assert a[3] == b[3] and a[6] == b[6] and a[7] == b[7]
    for k in (1, 5):
        for f in ("selected", "nearest", "res_last", "normvec"):
            if f in a[k]:
                np.testing.assert_array_equal(a[k][f], b[k][f], err_msg=f)
    assert a[0]["M"] == b[0]["M"] and a[4]["M"] == b[4]["M"]
    np.testing.assert_array_equal(a[0]["HtRinvH"], b[0]["HtRinvH"])
    np.testing.assert_array_equal(a[4]["HtRinvH"], b[4]["HtRinvH"])
    np.testing.assert_array_equal(a[2]["state"], b[2]["state"])


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [1, 5])
def test_gpu_cached_directory_probes_change_nothing(capi, scenes, cfg):
    """MALIO_OPT_PROBE_CACHE (default on): a search pass reuses the level-1 directory probe of the point's last search
    pass while the point stays in its cell. Same lists, same neighbours, same bits - across passes at moving states, a
    state that moves every point into another cell, a map that changes under the scan, and a new scan."""
    sc = scenes.make_scene(cfg=cfg)
    rng = np.random.default_rng(5)
    states = [np.array(sc["state0"], np.float64)]
    for k in range(3):
        st = states[0].copy()
        st[:3] += rng.normal(0, 0.01, 3)
        states.append(st)
    far = states[0].copy()
    far[:3] += (1.7, -2.3, 0.4)   # every point lands in another cell: every cached probe must be refused
    states.append(far)
    states.append(states[1])
    new = sc["map"][rng.integers(0, sc["map"].shape[0], 3000)].copy()
    new[:, :3] += rng.normal(0, 0.3, size=(3000, 3)).astype(np.float32)
    res = []
    for on in (1, 0):
        eng = capi.Engine(sc["params"])
        eng.set_option("probe_cache", on)
        eng.map_build(sc["map"])
        eng.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"])
        out = []
        for st in states:
            m = eng.measure(st, True)
            g = eng.scan_get()
            out.append((m["M"], m["HtRinvH"].copy(), g["nearest"].copy(), g["selected"].copy(), g["res_last"].copy()))
        eng.map_add(new, True)                       # lists appended to in place (some move): the cache must not survive
        m = eng.measure(states[1], True)
        g = eng.scan_get()
        out.append((m["M"], m["HtRinvH"].copy(), g["nearest"].copy(), g["selected"].copy(), g["res_last"].copy()))
        u = eng.update_iterated(sc["state0"], sc["P0"])
        eng.scan_set(sc["scan"][::2].copy(), sc["tables"], sc["temporal_comp"])   # a new scan: other points in the slots
        m = eng.measure(states[2], True)
        g = eng.scan_get()
        out.append((m["M"], m["HtRinvH"].copy(), g["nearest"].copy(), g["selected"].copy(), g["res_last"].copy()))
        res.append((out, u["state"].copy()))
    (a, ua), (b, ub) = res
    assert len(a) == len(b)
    for k, (x, y) in enumerate(zip(a, b)):
        assert x[0] == y[0], k
        for i in range(1, 5):
            np.testing.assert_array_equal(x[i], y[i], err_msg="step %d field %d" % (k, i))
    np.testing.assert_array_equal(ua, ub)


@pytest.mark.gpu
def test_gpu_ordered_lists_on_a_dense_and_on_a_tiny_map(capi, scenes):
    """Lists above 256 entries stay unordered and unflagged (walked whole); a map of a handful of points has lists of one or two
    entries; neither may change a neighbour. Same answers with MALIO_OPT_NL_SORTED on and off."""
    sc = scenes.make_scene(cfg=1)
    rng = np.random.default_rng(3)
    dense = sc["map"][:20000].copy()
    c = sc["map"][rng.integers(0, sc["map"].shape[0]), :3]
    dense[:, :3] = (c + rng.uniform(-2.5, 2.5, size=(20000, 3))).astype(np.float32)   # 160 points per m^3: lists of ~2 000 entries
    big = np.concatenate([sc["map"], dense])
    tiny = sc["map"][:7].copy()
    qs = np.concatenate([dense[:2000, :6] * 0 + dense[:2000, :6], sc["map"][:2000, :6]]).astype(np.float32)
    for mp, expect_unflagged in ((big, True), (tiny, False)):
        res = []
        for srt in (1, 0):
            eng = capi.Engine(sc["params"])
            eng.set_option("nl_sorted", srt)
            eng.map_build(mp)
            lo = eng.list_order()
            assert lo["broken"] == 0
            if srt:
                assert (lo["ordered"] < lo["lists"]) if expect_unflagged else (lo["ordered"] == lo["lists"]), lo
            eng.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"])
            m = eng.measure(sc["state0"], True)
            g = eng.scan_get()
            res.append((m["M"], g["nearest"].copy(), g["selected"].copy(), g["res_last"].copy()))
        assert res[0][0] == res[1][0]
        for k in (1, 2, 3):
            np.testing.assert_array_equal(res[0][k], res[1][k])


@pytest.mark.gpu
def test_gpu_cell_ordered_map_array_keeps_insertion_order_and_results(capi, scenes):
    """MALIO_OPT_MAP_CELL_ORDER (round 6, default on): a (re)build sorts the map array by level-1 cell, appends go behind it,
    d_map_ord remembers every slot's rank in INSERTION order. What must not change: malio_map_get hands the map out in insertion
    order - through a build, in-place appends, box deletions, and the rebuild that sweeps a fifth of the map's tombstones out and
    sorts again (ranks re-ranked, appended slots folded in) -, and a search finds the same neighbours whether the option is on or
    off and whether the map was handed over in raster or in shuffled order (exact ties aside: conftest.exact_ties)."""
    from conftest import exact_ties
    sc = scenes.make_scene(seed=641, N=8000, Nmap=120000, L=3)
    rng = np.random.default_rng(5)
    shuf = sc["map"][rng.permutation(sc["Nmap"])]
    ref = None
    for mp in (sc["map"], shuf):
        for co in (1, 0):
            e = capi.Engine(sc["params"])
            e.set_option("map_cell_order", co)
            e.map_build(mp)
            np.testing.assert_array_equal(e.map_get(), mp)  # insertion order, bit for bit, whatever the slots' order
            e.scan_set(sc["scan"], sc["tables"], sc["temporal_comp"])
            m = e.measure(sc["state0"], True)
            g = e.scan_get()
            if ref is None:
                ref = (m, g)
                continue
            ok = ~exact_ties(g, ref[1])
            assert m["M"] == ref[0]["M"] and np.array_equal(g["selected"][ok], ref[1]["selected"][ok])
            assert np.array_equal(g["normvec"][ok], ref[1]["normvec"][ok]) and np.array_equal(g["res_last"][ok], ref[1]["res_last"][ok])
            assert np.abs(m["HtRinvH"] - ref[0]["HtRinvH"]).max() <= 1e-11 * np.abs(ref[0]["HtRinvH"]).max()
    # a history of mutations on the shuffled map, cell order on: the expected map is kept on the host in insertion order
    e = capi.Engine(sc["params"])
    e.map_build(shuf)
    want = shuf.copy()
    key = lambda p: p[:, [0, 1, 2, 5]]
    for rnd in range(3):
        new = shuf[rng.integers(0, len(shuf), 1500)].copy()
        new[:, :3] += np.float32(300.0) + rng.normal(0, 5.0, (1500, 3)).astype(np.float32)  # empty space: all of them are kept, in order
        assert e.map_add(new, False) == 0
        want = np.concatenate([want, new])
        c = want[rng.integers(0, len(want)), :3]
        box = np.concatenate([c - 3.0, c + 3.0]).astype(np.float32)[None]
        inside = ((want[:, :3] >= box[0, :3]) & (want[:, :3] < box[0, 3:])).all(1)
        assert e.map_delete_boxes(box) == int(inside.sum())
        want = want[~inside]
        np.testing.assert_array_equal(key(e.map_get()), key(want))
    before = e.debug_counters()["rebuilds"]
    zc = np.median(want[:, 2])
    big = np.array([[-1e5, -1e5, -1e5, 1e5, 1e5, zc]], np.float32)  # half the map dies: the next search compacts, re-ranks and sorts again
    inside = ((want[:, :3] >= big[0, :3]) & (want[:, :3] < big[0, 3:])).all(1)
    assert e.map_delete_boxes(big) == int(inside.sum()) and inside.mean() > 0.25
    want = want[~inside]
    e.nearest_search(want[:100], 5)
    assert e.debug_counters()["rebuilds"] > before
    np.testing.assert_array_equal(key(e.map_get()), key(want))
    more = want[rng.integers(0, len(want), 500)].copy()
    more[:, :3] += np.float32(-300.0)
    assert e.map_add(more, False) == 0
    want = np.concatenate([want, more])
    np.testing.assert_array_equal(key(e.map_get()), key(want))
    fresh = capi.Engine(sc["params"])
    fresh.map_build(want)
    q = want[rng.integers(0, len(want), 3000)].copy()
    q[:, :3] += rng.normal(0, 0.3, (3000, 3)).astype(np.float32)
    _, d2a, ca = e.nearest_search(q, 5)
    _, d2b, cb = fresh.nearest_search(q, 5)
    np.testing.assert_array_equal(d2a, d2b)
    np.testing.assert_array_equal(ca, cb)
