"""conftest.exact_ties is what lets the parity tests tolerate an exact tie of float distances - so it must
tolerate nothing else. CPU only: identical searches, a swapped pair at EQUAL distance, a different fifth point at the same
distance (accepted), a different point at another distance, more tied queries than the limit, different counts (refused)."""
import numpy as np
import pytest

from conftest import exact_ties


def _pair(n=50):
    rng = np.random.default_rng(0)
    w = rng.integers(-8, 8, (n, 3)).astype(np.float32)           # small integers: every distance below is exact
    near = np.zeros((n, 5, 12), np.float32)
    off = np.array([[1, 0, 0], [0, 2, 0], [0, 0, 3], [4, 0, 0], [0, 5, 0]], np.float32)  # d2 = 1, 4, 9, 16, 25
    near[:, :, :3] = w[:, None, :] + off[None, :, :]
    mk = lambda a: dict(world=w.copy(), nearest=a.copy(), nearest_cnt=np.full(n, 5, np.int32))
    return mk(near), mk(near)


def test_identical_searches_have_no_ties():
    g, o = _pair()
    assert not exact_ties(g, o).any()


def test_a_swapped_pair_at_equal_distance_is_a_tie():
    g, o = _pair()
    # query 7: neighbours 0 and 1 both at d2 = 1 (+x and -x), returned in opposite orders
    for s in (g, o):
        s["nearest"][7, 1, :3] = s["world"][7] + np.array([-1, 0, 0], np.float32)
    g["nearest"][7, [0, 1]] = g["nearest"][7, [1, 0]]
    m = exact_ties(g, o)
    assert m.sum() == 1 and m[7]


def test_another_fifth_point_at_the_same_distance_is_a_tie():
    g, o = _pair()
    g["nearest"][3, 4, :3] = g["world"][3] + np.array([0, -5, 0], np.float32)  # d2 = 25 as well: the fifth against the sixth
    m = exact_ties(g, o)
    assert m.sum() == 1 and m[3]


def test_a_point_at_another_distance_is_refused():
    g, o = _pair()
    g["nearest"][3, 4, :3] = g["world"][3] + np.array([0, -5, 1], np.float32)  # d2 = 26
    with pytest.raises(AssertionError):
        exact_ties(g, o)
    g, o = _pair()
    g["nearest"][9, 2, 0] += np.float32(2.0 ** -10)  # a last-bits difference of one coordinate is NOT a tie
    with pytest.raises(AssertionError):
        exact_ties(g, o)


def test_too_many_ties_and_different_counts_are_refused():
    g, o = _pair()
    for q in range(9):  # nine swapped pairs: more than a handful in one pass means something else is going on
        for s in (g, o):
            s["nearest"][q, 1, :3] = s["world"][q] + np.array([-1, 0, 0], np.float32)
        g["nearest"][q, [0, 1]] = g["nearest"][q, [1, 0]]
    with pytest.raises(AssertionError):
        exact_ties(g, o)
    assert exact_ties(g, o, limit=9).sum() == 9
    g, o = _pair()
    g["nearest"][5, 4, :3] = 0
    g["nearest_cnt"][5] = 4
    with pytest.raises(AssertionError):
        exact_ties(g, o)
