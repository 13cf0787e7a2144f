"""Host-side pieces of the iterated update that the round-2 loop modes share (no GPU): the localization weight
(closed-form eigenvalues with a Jacobi fallback and bound-based short cuts, host/manifold.hpp) against NumPy's SVD, and
malio_ieskf_step - LU inversions with the substitution rows held in registers (host/ieskf.cpp) - against the oracle's
restatement of esekfom.hpp:521-720 and against a dense NumPy evaluation of the same formulas."""
import ctypes as C

import numpy as np
import pytest


def weight_ref(N, tmin, tmax, cmin, cmax):
    """laserMapping.cpp:745-756 on an M x 3 block whose Gram matrix is N."""
    ev = np.linalg.eigvalsh(N)
    w = np.sqrt(max(ev[0], 0.0)) / np.sqrt(ev[2])
    if w > tmax:
        return cmax
    if w < tmin:
        return cmin
    return (cmax - cmin) * (w - tmin) / (tmax - tmin) + cmin


def test_localize_weight_vs_svd(capi):
    rng = np.random.default_rng(3)
    f = capi.lib().malio_localize_weight
    tmin, tmax, cmin, cmax = 0.2, 0.7, 0.3, 2.0
    worst, n_lin = 0.0, 0
    for trial in range(4000):
        k = rng.integers(0, 4)
        H = rng.normal(size=(40, 3)) * np.array([1.0, 10.0 ** -rng.uniform(0, k), 10.0 ** -rng.uniform(0, k)])
        if trial % 3 == 0:
            H = H @ np.linalg.qr(rng.normal(size=(3, 3)))[0]          # weak direction off the axes
        if trial % 50 == 0:
            H[:, 1] = H[:, 0] * (1 + 1e-9 * rng.normal(size=40))       # two nearly coinciding eigenvalues
        if trial % 97 == 0:
            H *= 1e6                                                   # scale invariance
        N = H.T @ H
        n6 = (C.c_double * 6)(N[0, 0], N[1, 1], N[2, 2], N[0, 1], N[0, 2], N[1, 2])
        got, want = f(n6, tmin, tmax, cmin, cmax), weight_ref(N, tmin, tmax, cmin, cmax)
        if cmin < want < cmax:
            n_lin += 1
        # at a threshold the two evaluations may fall on different sides by rounding: the map is continuous at thresh_max
        # only up to the clamp value, so compare away from the two jumps
        sv = np.linalg.svd(H, compute_uv=False)
        w = sv[2] / sv[0]
        if min(abs(w - tmin), abs(w - tmax)) < 1e-9:
            continue
        worst = max(worst, abs(got - want) / want)
    assert n_lin > 300          # the linear branch (the one that needs the eigenvalues) was exercised
    assert worst < 1e-10


@pytest.mark.parametrize("L", [1, 3, 4])
def test_host_step_vs_dense_numpy(capi, scenes, L):
    """One iteration of esekfom.hpp:621-642 at the prior (dx = 0, so the projections are the identity up to the S2 block):
    K_h, K_x and dx_ against a dense NumPy evaluation of P_inv = (P^-1 + blk(HtRinvH))^-1."""
    sc = scenes.make_scene(seed=31, N=600, Nmap=6000, L=L)
    n, Cc = 17 + 6 * L, 6 * (L + 1)
    rng = np.random.default_rng(L)
    A = rng.normal(size=(Cc, 300))
    H = A @ A.T * 50.0
    h = rng.normal(size=Cc) * 10.0
    P0 = np.ascontiguousarray(sc["P0"], np.float64)
    x = capi.state_from_flat(sc["state0"], L)
    xp = capi.state_from_flat(sc["state0"], L)
    Pout = np.zeros((n, n))
    t, cv, dn = C.c_int(0), C.c_int(0), C.c_int(0)
    rc = capi.lib().malio_ieskf_step(L, 3, C.c_double(0.0), 0, C.byref(x), C.byref(xp), capi._p(P0, C.c_double),
                                     capi._p(np.ascontiguousarray(H), C.c_double), capi._p(h, C.c_double), C.byref(t),
                                     C.byref(cv), C.byref(dn), capi._p(Pout, C.c_double))
    assert rc == 0 and dn.value == 0
    Pinv = np.linalg.inv(P0)
    Pinv[:Cc, :Cc] += H
    Pi = np.linalg.inv(Pinv)
    dx = Pi[:, :Cc] @ h                      # K_h (dx_new = 0 at the prior)
    got = capi.state_to_flat(x, L) - sc["state0"]
    # position / extrinsic translation / vel / bg / ba blocks are plain additions: compare them with the dense solve
    assert np.allclose(got[0:3], dx[0:3], rtol=1e-7, atol=1e-12 * np.abs(dx).max())
    off = 6 + 3 * L
    sl = slice(7 + 4 * L, 7 + 7 * L)          # offset_T in the flat state (pos 3, rot 4, L quats, L translations, ...)
    assert np.allclose(got[sl], dx[off:off + 3 * L], rtol=1e-7, atol=1e-12 * np.abs(dx).max())
