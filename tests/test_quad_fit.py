"""The plane fit by four lanes (ma-lio_amd/csrc/quad_fit.hpp: the kernel's phase C since round 6) is ONE source for the device
and for the host; here the host instantiation - four lanes in lockstep - meets the oracle's esti_plane restatement
(oracle/orc_geom.cpp:16-156, common_lib.h:144-190) on millions of five-point sets, BIT FOR BIT: the planes scenes produce, and
the ones they rarely do - collinear and coincident points, a zero column, equal column norms (pivot ties), tiny and huge
coordinates - so that every select of the lane-parallel form has been on both sides. CPU only."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness(orc, tmp_path_factory):
    out = str(tmp_path_factory.mktemp("quadfit") / "libquadfit.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-o", out,
                           os.path.join(ROOT, "tests", "cpp", "quad_fit_host.cpp")])
    lib = C.CDLL(out)
    lib.quad_fit_compare.restype = C.c_long
    fn = C.cast(orc.lib().orc_esti_plane, C.c_void_p)

    def run(pts, threshold=0.4, wny=None, cov_threshold=0.5):
        pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 5, 3)
        first = C.c_long(-1)
        q4, o4, ok2 = (C.c_float * 4)(), (C.c_float * 4)(), (C.c_int * 2)()
        if wny is not None:
            wny = np.ascontiguousarray(wny, np.float32).reshape(-1, 5)
            assert wny.shape[0] == pts.shape[0]
        bad = lib.quad_fit_compare(pts.ctypes.data_as(C.POINTER(C.c_float)), C.c_long(pts.shape[0]), C.c_float(threshold), fn,
                                   C.byref(first), q4, o4, ok2, wny.ctypes.data_as(C.POINTER(C.c_float)) if wny is not None else None,
                                   C.c_double(cov_threshold))
        assert bad == 0, (bad, first.value, pts[first.value].tolist(), list(q4), list(o4), list(ok2))
        return pts.shape[0]
    return run


def _patches(rng, n, origin_scale):
    """five neighbours as a scan finds them: a 0.5 m-voxel patch of a plane of random orientation, 2 cm noise, somewhere in a map"""
    c = rng.uniform(-origin_scale, origin_scale, (n, 1, 3))
    nrm = rng.normal(size=(n, 3))
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    u = np.cross(nrm, rng.normal(size=(n, 3)))
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    v = np.cross(nrm, u)
    ab = rng.uniform(-0.7, 0.7, (n, 5, 2))
    return c + ab[:, :, :1] * u[:, None, :] + ab[:, :, 1:] * v[:, None, :] + rng.normal(0, 0.02, (n, 5, 1)) * nrm[:, None, :]


def test_scene_like_patches(harness):
    rng = np.random.default_rng(1)
    total = 0
    for scale in (1.0, 30.0, 300.0, 1000.0, 20000.0):
        total += harness(_patches(rng, 400_000, scale))
    assert total == 2_000_000


def test_axis_aligned_surfaces_and_pivot_ties(harness):
    """floors and walls of the synthetic scenes (one coordinate nearly constant), symmetric sets whose column norms tie exactly"""
    rng = np.random.default_rng(2)
    p = rng.uniform(-50, 50, (300_000, 5, 3))
    ax = rng.integers(0, 3, 300_000)
    val = rng.choice([0.0, -1.8, 6.0, 40.0, -60.0, 1e-3], 300_000)
    p[np.arange(300_000), :, ax] = val[:, None] + rng.normal(0, 0.02, (300_000, 5)) * (rng.random((300_000, 1)) < 0.7)
    harness(p)
    q = rng.integers(-3, 4, (300_000, 5, 3)).astype(np.float64)  # small integers: equal norms, exact cancellations, zero columns
    harness(q)
    harness(q * 0.25 + np.array([1.0, 1.0, 1.0]))
    s = rng.uniform(-2, 2, (100_000, 5, 3))
    s[:, :, 1] = s[:, :, 0]  # two identical columns: rank 2, a tie at every pivot
    harness(s)
    s[:, :, 2] = -s[:, :, 0]
    harness(s)


def test_degenerate_sets(harness):
    rng = np.random.default_rng(3)
    n = 100_000
    t = rng.uniform(-1, 1, (n, 5, 1))
    d = rng.normal(size=(n, 1, 3))
    line = rng.uniform(-20, 20, (n, 1, 3)) + t * d  # collinear
    harness(line)
    harness(np.repeat(rng.uniform(-20, 20, (n, 1, 3)), 5, axis=1))  # five times the same point
    z = rng.uniform(-5, 5, (n, 5, 3))
    z[:, :, rng.integers(0, 3)] = 0.0  # a zero column
    harness(z)
    harness(np.zeros((10, 5, 3)))
    two = rng.uniform(-5, 5, (n, 5, 3))
    two[:, 2:, :] = two[:, 1:2, :]  # two distinct points only
    harness(two)


def test_extreme_magnitudes(harness):
    rng = np.random.default_rng(4)
    n = 100_000
    for e in (-30, -20, -12, -6, 6, 12, 18):
        harness(rng.uniform(-1, 1, (n, 5, 3)) * 10.0 ** e)
    mix = rng.uniform(-1, 1, (n, 5, 3)) * 10.0 ** rng.integers(-20, 15, (n, 1, 3))  # columns of very different scale
    harness(mix)
    rows = rng.uniform(-1, 1, (n, 5, 3)) * 10.0 ** rng.integers(-25, 10, (n, 5, 1))  # rows of very different scale: flat tails
    harness(rows)


def test_plane_covariance_by_four_lanes(harness):
    """esti_plane's plane_cov (common_lib.h:159-173): the weights of the five neighbours' normal_y, bit for bit - uniform maps
    (every normal_y 0.001), BASELINE config 3's U[0, 0.002], values around the threshold, a first neighbour at or below 1e-5"""
    rng = np.random.default_rng(5)
    n = 300_000
    p = _patches(rng, n, 30.0)
    harness(p, wny=np.full((n, 5), 0.001))
    harness(p, wny=rng.uniform(0, 0.002, (n, 5)))
    harness(p, wny=rng.uniform(0, 1.0, (n, 5)))
    w = rng.uniform(0, 0.002, (n, 5))
    w[:, 0] = rng.choice([0.0, 1e-5, 9.9e-6, 1.1e-5, 1e-6], n)
    harness(p, wny=w)
    harness(p, wny=np.full((n, 5), 0.5))  # every |tau - W| zero: 0 / 0
