"""a15 (host): the product's Barfoot compounding / point covariance (libmalio_hip.so, pure host code) against
the oracle restatement of associate_uct.hpp, including the reference's aliased call pattern."""
import numpy as np


def rnd_pose(scenes, rng, scale):
    A = rng.normal(size=(6, 6))
    return scenes.make_pose(scenes.q_from_rotvec(rng.normal(size=3) * 0.6), rng.normal(size=3) * 2, scale * (A @ A.T))


def test_compound_matches_oracle(capi, orc, scenes):
    rng = np.random.default_rng(17)
    for scale in (1e-8, 1e-4, 1e-2):
        for _ in range(20):
            p1, p2 = rnd_pose(scenes, rng, scale), rnd_pose(scenes, rng, scale)
            for inverse in (False, True):
                for alias in (False, True):
                    got = capi.compound(p1, p2, inverse, alias)
                    want = orc.compound(p1, p2, inverse, alias)
                    assert np.allclose(got[:23], want[:23], rtol=0, atol=1e-13)
                    assert np.allclose(got[23:], want[23:], rtol=1e-11, atol=1e-20)


def test_table_chain_like_reference(capi, orc, scenes):
    """laserMapping.cpp:1042-1044: comp(ext, entry) -> comp(tc, .) aliased -> invcomp(ext0, .) aliased."""
    rng = np.random.default_rng(19)
    ext, ext0, tc, entry = (rnd_pose(scenes, rng, 1e-6) for _ in range(4))
    g = capi.compound(ext, entry)
    g = capi.compound(tc, g, alias=True)
    g = capi.compound(ext0, g, inverse=True, alias=True)
    w = orc.compound(ext, entry)
    w = orc.compound(tc, w, alias=True)
    w = orc.compound(ext0, w, inverse=True, alias=True)
    assert np.allclose(g, w, rtol=1e-11, atol=1e-18)


def test_point_uncertainty_matches_oracle_and_trace_fold(capi, orc, scenes):
    rng = np.random.default_rng(23)
    for _ in range(50):
        pose = rnd_pose(scenes, rng, 1e-5)
        p = np.zeros(12, np.float32)
        p[:3] = rng.uniform(-90, 90, 3)
        got = capi.eval_point_uncertainty(p, pose)
        want = orc.eval_point_uncertainty(p, pose)
        assert np.allclose(got, want, rtol=1e-12, atol=1e-16)
