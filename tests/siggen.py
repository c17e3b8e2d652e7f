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
