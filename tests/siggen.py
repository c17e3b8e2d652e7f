"""Seeded synthetic inputs for the parity tests and the bench."""
import numpy as np


def awgn(rng, n, power):
    s = np.sqrt(power / 2.0)
    return (rng.standard_normal(n) * s + 1j * rng.standard_normal(n) * s).astype(np.complex64)


def loop_to(x, n, rotate=0):
    """Loop one radio frame to n samples, starting `rotate` samples into the frame."""
    reps = (n + rotate) // x.size + 2
    return np.tile(x, reps)[rotate:rotate + n]


def apply_cfo(x, cfo_subcarriers, fft=128):
    """Rotate by cfo (in units of the 15 kHz subcarrier spacing) at 1.92 Msps."""
    n = np.arange(x.size, dtype=np.float64)
    return (x * np.exp(2j * np.pi * cfo_subcarriers * n / fft)).astype(np.complex64)


def scan_channels(base_frames, n_channels, n_samples, snr_db=20.0, cfo=False, seed0=0):
    """Channel c: base = base_frames[c % len], rotated by a seeded offset, looped, optional CFO,
    AWGN at snr_db; every 8th channel (c % 8 == 7) is noise only.
    base_frames: list of (complex64 one-frame array at 1.92 Msps (19200), expected cell_id).
    Returns iq [n_channels, n_samples] complex64 and expected cell ids (-1 = none)."""
    iq = np.empty((n_channels, n_samples), np.complex64)
    expect = np.full(n_channels, -1, np.int64)
    for c in range(n_channels):
        rng = np.random.default_rng(seed0 + c)
        frame, cid = base_frames[c % len(base_frames)]
        p_sig = float(np.mean(np.abs(frame) ** 2))
        rot = int(rng.integers(0, frame.size))
        f = float(rng.uniform(-0.2, 0.2)) if cfo else 0.0
        noise = awgn(rng, n_samples, p_sig / (10.0 ** (snr_db / 10.0)))
        if c % 8 == 7:
            iq[c] = awgn(rng, n_samples, p_sig)
            continue
        s = loop_to(frame, n_samples, rot)
        if f:
            s = apply_cfo(s, f)
        iq[c] = s + noise
        expect[c] = cid
    return iq, expect


def random_scene(seed, frames_192, names, min_windows=25, max_windows=45, snr_range=(-2.0, 25.0)):
    """A randomised batch for the record-parity tests (tests/test_gpu_parity.py, tools/stress_parity.py): per
    channel a random fixture (or none), rotation, SNR between -2 and 25 dB, CFO, a random dropout or level step, a
    capture length that is not a multiple of anything, an odd channel count; random threshold and tracking
    parameters.  Returns (iq [n_ch, n] complex64, keyword arguments of the batch)."""
    rng = np.random.default_rng(seed)
    n_ch = int(rng.integers(3, 12))
    n = int(rng.integers(min_windows, max_windows)) * 9600 + int(rng.integers(0, 9600))
    iq = np.empty((n_ch, n), np.complex64)
    for c in range(n_ch):
        kind = int(rng.integers(0, 5))
        if kind == 0:
            iq[c] = awgn(rng, n, 1.0)
            continue
        frame = frames_192[names[int(rng.integers(0, len(names)))]][19200:38400]
        p_sig = float(np.mean(np.abs(frame) ** 2))
        s = loop_to(frame, n, int(rng.integers(0, frame.size)))
        s = apply_cfo(s, float(rng.uniform(-0.2, 0.2)))
        snr = float(rng.uniform(*snr_range))
        x = s + awgn(rng, n, p_sig / 10.0 ** (snr / 10.0))
        if kind == 2:  # dropout
            a = int(rng.integers(0, n // 2))
            m = min(n - a, int(rng.integers(9600, 20 * 9600)))
            x[a:a + m] = awgn(rng, m, p_sig * 0.05)
        if kind == 3:  # level step
            a = int(rng.integers(0, n))
            x[a:] *= np.float32(rng.uniform(0.05, 20.0))
        if kind == 4:  # signal disappears
            a = int(rng.integers(n // 4, n))
            x[a:] = awgn(rng, n - a, p_sig * 0.05)
        iq[c] = x
    kw = dict(psr_threshold=float(rng.choice([2.5, 4.0, 6.0])), track_after=int(rng.choice([2, 8, 16])),
              track_every=int(rng.choice([0, 3, 8])))
    return iq, kw
