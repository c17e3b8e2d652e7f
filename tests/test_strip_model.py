"""The strip decomposition of srslte_pss_find_pss's search tail (tests/strip_model.py, the numpy twin of
csrc/lt_scan.h) against the direct algorithm (oracle/lte_oracle.c:295-316) on random and adversarial EMA vectors."""
import numpy as np
import pytest

import strip_model as M


def _vec(rng, kind):
    a = np.zeros(M.CONV_LEN + 1, np.float32)
    if kind == "exp":      # |complex gaussian|^2, the metric of noise
        a[:M.NLAGS] = rng.exponential(1.0, M.NLAGS)
    elif kind == "smooth":  # band-limited noise: neighbouring lags correlated, lobes 2-6 lags wide
        x = rng.normal(size=M.NLAGS + 8) + 1j * rng.normal(size=M.NLAGS + 8)
        k = np.hanning(7)
        y = np.convolve(x, k, "same")[:M.NLAGS]
        a[:M.NLAGS] = np.abs(y) ** 2
    elif kind == "gamma":   # EMA of several windows
        a[:M.NLAGS] = rng.gamma(5.0, 1.0, M.NLAGS)
    return a.astype(np.float32)


def _check(a):
    """Both decompositions -- the strips' own lobe walks (per-chain kernels) and the deferred form (scan rounds: heads and
    rows, the resolver walks the one lobe that matters) -- against the direct algorithm."""
    want = M.find_reference(a)
    ok = True
    for find in (M.find_strips, M.find_strips_deferred):
        got = find(a)
        if got is None:
            ok = False
            continue
        assert got[0] == want[0], (find.__name__, got, want)
        assert got[1] == want[1]
        assert (got[2] == want[2]) or (np.isnan(got[2]) and np.isnan(want[2])), (find.__name__, got, want)
    return ok


@pytest.mark.parametrize("kind", ["exp", "smooth", "gamma"])
def test_random_vectors(kind):
    rng = np.random.default_rng(7)
    resolved = sum(_check(_vec(rng, kind)) for _ in range(40))
    assert resolved == 40


def test_peak_at_every_interesting_lag():
    """A strong peak with a lobe of 1-5 lags planted at the window ends, at lags 0-3, and on / next to every
    strip boundary, on top of correlated noise."""
    rng = np.random.default_rng(11)
    spots = [0, 1, 2, 3, 4, 30, 31, 32, 33, M.NLAGS - 1, M.NLAGS - 2, M.NLAGS - 3, M.NLAGS - 31, M.NLAGS - 33]
    for s in range(1, M.STRIPS):
        b = M.PITCH * s
        spots += [b - 3, b - 2, b - 1, b, b + 1, b + 2, b - M.HALO, b + M.HALO]
    n_res = 0
    for p in spots:
        for width in (1, 2, 3, 5):
            a = _vec(rng, "smooth")
            top = a.max() * 3
            for d in range(-width, width + 1):
                if 0 <= p + d < M.NLAGS:
                    a[p + d] = top * (1.0 - 0.15 * abs(d))
            n_res += _check(a)
    assert n_res == len(spots) * 4


def test_second_peak_next_to_a_boundary():
    """The side lobe that sets the PSR sits inside the neighbouring strip, right after the main lobe."""
    rng = np.random.default_rng(5)
    for s in range(1, M.STRIPS):
        b = M.PITCH * s
        for dp, ds in ((-2, 3), (1, -4), (-1, 2), (0, -3)):
            a = _vec(rng, "gamma")
            a[b + dp] = 100.0
            a[b + dp - 1] = 60.0
            a[b + dp + 1] = 60.0
            a[b + ds] = max(a[b + ds], 80.0) if abs(ds - dp) > 1 else a[b + ds]
            assert _check(a)


def test_wide_lobe_is_flagged_not_wrong():
    a = np.zeros(M.CONV_LEN + 1, np.float32)
    a[:M.NLAGS] = 1.0 + 0.001 * np.random.default_rng(3).random(M.NLAGS)
    p = 4000
    for d in range(-60, 61):
        a[p + d] = 50.0 - 0.5 * abs(d)
    assert M.find_strips(a) is None
    assert M.find_strips_deferred(a) is None


def test_degenerate_vectors_do_not_crash():
    for a in (np.zeros(M.CONV_LEN + 1, np.float32), np.ones(M.CONV_LEN + 1, np.float32)):
        a[M.NLAGS:] = 0
        for find in (M.find_strips, M.find_strips_deferred):
            got = find(a)
            if got is not None:
                want = M.find_reference(a)
                assert got[0] == want[0]
