"""Window cache level 4 (lt_batch_set_window_cache): the per-chain kernel's searches out of the chain's lag-metric STREAM.

Levels 0..3 give the same bits among themselves (tests/test_gpu_parity.py); level 4 forms the interior lags of a window
from overlap-save blocks laid on the chain's stream instead of on the window, and the 253 edge lags by direct sums: the
same sums in another order of rounding.  Its bar is the one every level has against the oracle -- integer, index and flag
fields equal, everything downstream of the lag metric bit-exact, the lag-metric quantities within 1e-4 relative -- and it
is held against the oracle AND against level 0 here.
"""
import numpy as np
import pytest

from conftest import FIXTURES
import siggen
from test_gpu_parity import assert_records_match

pytestmark = pytest.mark.gpu


def run_level(lt, iq, level, mode=1, runs=2, **kw):
    """Records of the last of `runs` identical runs (levels 3 and 4 give a fresh batch's first run one pass: the second is
    the one with the cache) and that run's cache counters."""
    b = lt.Batch(iq.shape[0], iq.shape[1], **kw)
    b.set_mode(mode)
    b.set_window_cache(level)
    got = None
    for _ in range(runs):
        b.reset()
        b.run_host(iq)
        got = [r.copy() for r in b.collect()]
    return got, b.cache_stats()


@pytest.mark.parametrize("mode", [1, 0])
def test_stream_gives_the_oracles_records_at_marginal_snr(lt, oracle, frames_192, mode):
    """Weak cells that keep firing and losing, every hit moving the chain's grid: the searches come out of the stream
    whatever grid the chain is on (none goes through the transforms once the stream is there), and the records are the
    oracle's and level 0's."""
    base = [(frames_192[n][19200:38400], FIXTURES[n][2]["cell_id"]) for n in FIXTURES]
    iq, _ = siggen.scan_channels(base, 6, 90 * 9600 + 41, snr_db=1.0, cfo=True, seed0=270)
    got4, stats = run_level(lt, iq, 4, mode)
    got0, _ = run_level(lt, iq, 0, mode, runs=1)
    assert stats[3] > 0 and stats[1] > 50, f"the scene must send chains through the stream: {stats}"
    assert stats[4] > 0, f"... and move their grids: {stats}"
    want = oracle.batch_run(iq, nthreads=8)
    for c in range(18):
        assert_records_match(got4[c], want[c], f"level 4 against the oracle, chain {c}")
        assert_records_match(got4[c], got0[c], f"level 4 against level 0, chain {c}")
    # what a chain with a stream still sends through the transforms: windows in front of the stream only
    _, stats3 = run_level(lt, iq, 3, mode)
    assert stats[2] < stats3[2] or stats3[2] == 0, f"fewer searches through the transforms than at level 3: {stats} vs {stats3}"


@pytest.mark.parametrize("lag", [0, 5, 63, 64, 126, 127, 128, 9598, 9599, 9600, 9601, 9663, 9664, 9700, 9725])
def test_stream_edge_lags(lt, oracle, frames_192, lag):
    """The lags whose sums stop at the window's first or last sample (0..126 and 9600..9725) are formed directly, the
    ones next to them come out of the stream: a capture built so that the strongest lag of the first windows is `lag` --
    a PSS cut off by the window's start, or reaching past its end -- against the oracle, peak position and value."""
    x = frames_192["50prb"][19200:38400]
    rng = np.random.default_rng(1000 + lag)
    n = 40 * 9600 + 17
    # where the PSS ends in the frame: the strongest lag of a window aligned with the frame
    o = oracle.Pss(FIXTURES["50prb"][2]["cell_id"] % 3)
    p0, _, _ = o.find_pss(x[:9600])
    # lag p0 -> lag `lag`: start the capture (p0 - lag) samples into the frame
    s = siggen.loop_to(x, n, (p0 - lag) % 9600)
    p_sig = float(np.mean(np.abs(x) ** 2))
    iq = np.stack([s + siggen.awgn(rng, n, p_sig / 10.0 ** (s_db / 10.0)) for s_db in (30.0, 3.0, 0.0)]).astype(np.complex64)
    # a threshold nothing reaches: the chains stay idle on the capture's grid, every search sees the peak at `lag`
    kw = dict(psr_threshold=1e9)
    got4, stats = run_level(lt, iq, 4, 0, **kw)
    assert stats[1] > 30 * 9, f"the searches must come out of the stream: {stats}"
    want = oracle.batch_run(iq, nthreads=8, **kw)
    for c in range(9):
        assert_records_match(got4[c], want[c], f"chain {c}")


@pytest.mark.parametrize("seed", [4100, 4101, 4102, 4103, 4104, 4105])
def test_stream_random_scenes(lt, oracle, frames_192, seed):
    """Random scenes of tools/stress_parity.py at marginal SNR, both modes, against the oracle."""
    iq, kw = siggen.random_scene(seed, frames_192, list(FIXTURES), 40, 90, (-6.0, 6.0))
    want = oracle.batch_run(iq, nthreads=8, **kw)
    for mode in (1, 0):
        got, stats = run_level(lt, iq, 4, mode, **kw)
        for c in range(len(want)):
            assert_records_match(got[c], want[c], f"seed {seed} mode {mode} chain {c}")


def test_stream_with_pushed_chunks(lt, oracle, frames_192):
    """lt_batch_push at level 4: a chain's stream starts in front of where the chain stands -- in the history the
    previous chunks left behind -- and ends with the chunk; the records of the chunks are those of the whole capture."""
    base = [(frames_192[n][19200:38400], FIXTURES[n][2]["cell_id"]) for n in FIXTURES]
    iq, _ = siggen.scan_channels(base, 6, 120 * 9600 + 977, snr_db=0.0, cfo=True, seed0=505)
    want = oracle.batch_run(iq, nthreads=8)
    chunk = 45 * 9600 + 123
    n = iq.shape[1]
    for mode in (1, 0):
        b = lt.Batch(6, chunk)
        b.set_mode(mode)
        b.set_window_cache(4)
        got = [[] for _ in range(18)]
        for a in range(0, n, chunk):
            e = min(n, a + chunk)
            b.push_host(iq[:, a:e], end_of_stream=(e == n))
            for c, r in enumerate(b.collect()):
                got[c].append(r.copy())
        stats = b.cache_stats()
        assert stats[1] > 0, f"the last chunk's searches must come out of the stream: {stats}"
        for c in range(18):
            assert_records_match(np.concatenate(got[c]), want[c], f"mode {mode} chain {c}")


def test_stream_shorter_than_the_capture(lt, oracle, frames_192):
    """A chain's stream holds 256 slots of 10 752 positions at most (2.75 M samples): in a longer capture the stream kernel
    puts the chain down where its stream ends and the second pass of the per-chain kernel takes it up, through the
    transforms -- the records are the oracle's all the same, and both kinds of search have happened."""
    base = [(frames_192[n][19200:38400], FIXTURES[n][2]["cell_id"]) for n in FIXTURES]
    iq, _ = siggen.scan_channels(base, 3, 330 * 9600 + 77, snr_db=-1.0, cfo=True, seed0=911)
    want = oracle.batch_run(iq, nthreads=8)
    for mode in (0, 1):
        got, stats = run_level(lt, iq, 4, mode)
        assert stats[1] > 200, f"searches out of the streams: {stats}"
        if mode == 0:
            assert stats[2] > 20, f"... and, behind their ends, through the transforms: {stats}"
        for c in range(9):
            assert_records_match(got[c], want[c], f"mode {mode} chain {c}")


@pytest.mark.parametrize("mode", [1, 0])
def test_level4_is_deterministic(lt, oracle, frames_192, mode):
    """The same capture five times at level 4 (the default): byte-identical records -- on a fresh batch each time, on one
    batch run again and again after lt_batch_reset, with the host far ahead of the device (nothing collected in between)
    and right behind it (collected after every run).  Round 4's level 4 took the one-pass / two-pass decision from a word
    of pinned host memory the kernel writes asynchronously and handed the cache's slots out first come first served:
    both could change the low bits of PSR from run to run.  50 channels = 150 chains at marginal SNR, the case in which the
    old pool ran short."""
    base = [(frames_192[n][19200:38400], FIXTURES[n][2]["cell_id"]) for n in FIXTURES]
    iq, _ = siggen.scan_channels(base, 50, 45 * 9600 + 333, snr_db=-1.0, cfo=True, seed0=7000)
    runs = []
    for k in range(3):  # fresh batches (the library's default level: 4)
        b = lt.Batch(50, iq.shape[1])
        b.set_mode(mode)
        b.run_host(iq)
        runs.append([r.copy() for r in b.collect()])
        stats = b.cache_stats()
        assert stats[3] > 20 and stats[1] > 500, f"the scene must send chains through their streams in the very first run: {stats}"
        b.close()
    b = lt.Batch(50, iq.shape[1])
    b.set_mode(mode)
    for k in range(3):  # one batch, collected after every run
        b.reset()
        b.run_host(iq)
        runs.append([r.copy() for r in b.collect()])
    for k in range(4):  # ... and with four runs enqueued before the host looks at anything
        b.reset()
        b.run_host(iq)
    runs.append([r.copy() for r in b.collect()])
    for k, other in enumerate(runs[1:]):
        for c in range(150):
            assert other[c].tobytes() == runs[0][c].tobytes(), f"run {k + 1} chain {c}"
    want = oracle.batch_run(iq, nthreads=8)
    for c in range(150):
        assert_records_match(runs[0][c], want[c], f"chain {c}")
