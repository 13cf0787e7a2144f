"""Voxel down-sampling: pcl::VoxelGrid<PointXYZINormal> as restated (PCL is not vendored with
the reference and not installed: parity with a PCL build is UNPINNED). CPU: the oracle restatement against an
independent NumPy group-by. GPU: malio_voxel_downsample against the oracle, bit for bit."""
import numpy as np
import pytest


def _cloud(rng, n, extent=40.0, origin=(13.0, -7.0, 2.0)):
    p = np.zeros((n, 12), np.float32)
    p[:, :3] = (rng.uniform(-extent, extent, (n, 3)) * np.array([1, 1, 0.1]) + np.array(origin)).astype(np.float32)
    p[:, 3] = 1.0
    p[:, 4] = rng.integers(0, 10, n)            # normal_x: uncertainty-table index after undistortion? no: see :973
    p[:, 5] = rng.uniform(0, 3, n)
    p[:, 6] = rng.normal(size=n)
    p[:, 8] = rng.integers(0, 10, n)            # intensity = idx written by the undistortion (IMU_Processing.hpp:504)
    p[:, 9] = rng.uniform(0, 100, n)            # curvature = time offset in ms
    return p


def _numpy_voxelgrid(p, leaf):
    """Independent formulation: float64 group means keyed by the integer voxel coordinates."""
    inv = np.float32(1.0) / np.float32(leaf)
    ijk = np.floor(p[:, :3] * inv).astype(np.int64)
    ijk -= ijk.min(0)
    div = ijk.max(0) + 1
    key = ijk[:, 0] + ijk[:, 1] * div[0] + ijk[:, 2] * div[0] * div[1]
    order = np.argsort(key, kind="stable")
    ks = key[order]
    starts = np.r_[0, np.nonzero(np.diff(ks))[0] + 1]
    cnt = np.diff(np.r_[starts, len(ks)])
    sums = np.add.reduceat(p[order].astype(np.float64), starts, axis=0)
    return sums / cnt[:, None], cnt


@pytest.mark.parametrize("seed,n,leaf", [(1, 20000, 0.5), (2, 5000, 0.3), (3, 300, 2.0), (4, 1, 0.5)])
def test_oracle_voxelgrid_matches_numpy_groupby(orc, seed, n, leaf):
    rng = np.random.default_rng(seed)
    p = _cloud(rng, n)
    out = orc.voxel_downsample(p, leaf, normalize_normal=False)
    mean, cnt = _numpy_voxelgrid(p, leaf)
    assert out.shape[0] == mean.shape[0]
    for col in (0, 1, 2, 4, 5, 6, 8, 9):   # every averaged field, output in ascending voxel-index order
        np.testing.assert_allclose(out[:, col], mean[:, col], rtol=2e-6, atol=2e-5)
    on = orc.voxel_downsample(p, leaf, normalize_normal=True)
    nrm = np.linalg.norm(on[:, 4:8].astype(np.float64), axis=1)
    assert np.allclose(nrm[nrm > 0], 1.0, atol=1e-6)
    np.testing.assert_array_equal(on[:, [0, 1, 2, 8, 9]], out[:, [0, 1, 2, 8, 9]])


def test_oracle_voxelgrid_edge_cases(orc):
    rng = np.random.default_rng(0)
    assert orc.voxel_downsample(np.zeros((0, 12), np.float32), 0.5).shape[0] == 0
    p = _cloud(rng, 100)
    p[::7, 0] = np.nan                                  # non-finite points are skipped
    out = orc.voxel_downsample(p, 0.5)
    assert np.isfinite(out[:, :3]).all() and 0 < out.shape[0] <= 100 - len(p[::7])
    q = _cloud(rng, 50, extent=1e6)                     # index overflow: PCL warns and returns the input unchanged
    np.testing.assert_array_equal(orc.voxel_downsample(q, 0.01), q)
    r = np.tile(_cloud(rng, 1), (9, 1))                 # nine copies of one point -> one voxel, the point itself
    o = orc.voxel_downsample(r, 0.5, normalize_normal=False)
    assert o.shape[0] == 1 and np.allclose(o[0, :3], r[0, :3], rtol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n,leaf,mode", [(1, 200000, 0.5, 1), (2, 30000, 0.3, 0), (3, 500, 2.0, 1), (4, 1, 0.5, 0),
                                              (5, 1_200_000, 0.5, 1)])  # > 1 M points: the three-kernel scan path
def test_gpu_voxelgrid_equals_oracle(orc, capi, scenes, seed, n, leaf, mode):
    rng = np.random.default_rng(seed)
    p = _cloud(rng, n)
    if n > 100:
        p[5::97, 1] = np.inf
    eng = capi.Engine(scenes.make_scene(cfg=1)["params"])
    got = eng.voxel_downsample(p, leaf, mode)
    want = orc.voxel_downsample(p, leaf, normalize_normal=bool(mode))
    assert got.shape == want.shape
    np.testing.assert_array_equal(got, want)


@pytest.mark.gpu
def test_gpu_voxelgrid_edge_cases(orc, capi, scenes):
    rng = np.random.default_rng(9)
    eng = capi.Engine(scenes.make_scene(cfg=1)["params"])
    assert eng.voxel_downsample(np.zeros((0, 12), np.float32), 0.5).shape[0] == 0
    q = _cloud(rng, 50, extent=1e6)
    np.testing.assert_array_equal(eng.voxel_downsample(q, 0.01), q)
    allnan = np.full((10, 12), np.nan, np.float32)
    assert eng.voxel_downsample(allnan, 0.5).shape[0] == 0
    with pytest.raises(RuntimeError):
        eng.voxel_downsample(_cloud(rng, 10), 0.0)
