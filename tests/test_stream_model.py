"""The decomposition window cache level 4 rests on (csrc/lt_chain_kernel.h, "lag-metric stream"), on the CPU against the
oracle: lag l of the window at w0 is the stream's value at position w0 + l for 127 <= l < 9600 -- whatever w0 is --, and a
truncated sum over the window's first or last samples for the 253 edge lags."""
import numpy as np
import pytest

from oracle import oracle as O


def _metric_of_window(root, win):
    """The oracle's lag metric of one window: a fresh srslte_pss_find_pss leaves avg = 0.2 * metric behind."""
    p = O.Pss(root)
    p.find_pss(win)
    return p.avg()[:9726].astype(np.float64) / 0.2


@pytest.mark.parametrize("root", [0, 1, 2])
def test_interior_lags_depend_on_the_position_only_and_edge_lags_are_truncated_sums(root):
    rng = np.random.default_rng(40 + root)
    n = 3 * 9600 + 500
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    h = O.pss_taps(root).astype(np.complex128)          # 128 taps
    xd = x.astype(np.complex128)
    full = np.convolve(xd, h)                            # full[a] = sum_k h[k] x[a - k]
    stream = np.abs(full) ** 2                           # the chain's stream, position by position
    scale = float(stream.max())
    for w0 in (0, 1, 77, 4801, 9600 + 333, 2 * 9600 + 499):
        win = x[w0:w0 + 9600]
        m = _metric_of_window(root, np.ascontiguousarray(win))
        # interior lags: every tap meets a sample of the window
        lag = np.arange(127, 9600)
        np.testing.assert_allclose(m[lag], stream[w0 + lag], rtol=0, atol=2e-5 * scale)
        # head lags 0..126: taps 0..l on samples l..0; tail lags 9600..9725: taps l - 9599..127 on the window's last samples
        wd = win.astype(np.complex128)
        for l in (0, 1, 63, 64, 126):
            s = sum(h[k] * wd[l - k] for k in range(l + 1))
            assert abs(m[l] - abs(s) ** 2) <= 2e-5 * scale
        for l in (9600, 9601, 9663, 9664, 9725):
            s = sum(h[k] * wd[l - k] for k in range(l - 9599, 128))
            assert abs(m[l] - abs(s) ** 2) <= 2e-5 * scale
        # ... and the edge lags are NOT the stream's values (the stream's sums run on into the neighbouring samples)
        if w0 >= 127:
            assert abs(m[0] - stream[w0]) > 1e-3 * scale or abs(m[1] - stream[w0 + 1]) > 1e-3 * scale
