"""CPU tests of the oracle: sequence generators against 3GPP TS 36.211 facts, the FFT-9728 against
numpy, and the pinning of the oracle on the reference's four fixtures
(python/qa_downlink_trigger_c.py:93-98,128-133,163-168,198-203)."""
import numpy as np
import pytest

from conftest import FIXTURES, load_frame


def test_zadoff_chu_properties(oracle):
    for r, u in enumerate((25, 29, 34)):
        d = oracle.pss_zc(r).astype(np.complex128)
        n = np.arange(62)
        ref = np.where(n < 31, np.exp(-1j * np.pi * u * n * (n + 1) / 63),
                       np.exp(-1j * np.pi * u * (n + 1) * (n + 2) / 63))
        assert np.allclose(d, ref, atol=1e-6)
        assert np.allclose(np.abs(d), 1.0, atol=1e-6)          # constant amplitude
        assert np.allclose(d, d[::-1], atol=1e-6)              # d[n] = d[61-n]
    # root 34 = 63 - 29: conjugate sequences
    assert np.allclose(oracle.pss_zc(2), np.conj(oracle.pss_zc(1)), atol=1e-6)


def test_pss_taps_are_idft_of_zc(oracle):
    for r in range(3):
        h = oracle.pss_taps(r).astype(np.complex128)
        d = oracle.pss_zc(r).astype(np.complex128)
        X = np.zeros(128, complex)
        X[128 - 31:] = d[:31]
        X[1:32] = d[31:]
        x = np.fft.ifft(X) * 128
        assert np.allclose(h, np.conj(x) / np.sqrt(128) / 62, atol=1e-8)
        assert np.allclose(h[1:], h[1:][::-1], atol=1e-8)      # even symmetry used by the kernel


def test_sss_m_sequences_and_table(oracle):
    s, c, z = oracle.sss_mseq()
    for seq in (s, c, z):
        assert set(np.unique(seq)) == {-1, 1}
        assert int(seq.sum()) == -1                            # m-sequence balance: 16 ones, 15 zeros
        ac = [int(np.dot(seq, np.roll(seq, k))) for k in range(1, 31)]
        assert ac == [-1] * 30                                 # two-valued autocorrelation
    # 36.211 Table 6.11.2.1-1 spot values
    for n_id_1, (m0, m1) in {0: (0, 1), 1: (1, 2), 29: (29, 30), 30: (0, 2), 59: (0, 3),
                             165: (0, 7), 166: (1, 8), 167: (2, 9)}.items():
        assert oracle.sss_m0m1_of(n_id_1) == (m0, m1)
    pairs = {oracle.sss_m0m1_of(n) for n in range(168)}
    assert len(pairs) == 168
    t = oracle.nid1_table()
    for n in range(168):
        m0, m1 = oracle.sss_m0m1_of(n)
        assert t[m0, m1 - 1] == n
    assert np.count_nonzero(t) == 167                          # N_id_1 = 0 is indistinguishable from "unset"
    sy = oracle.Sync(0)
    assert sy.n_id_1(0, 0) == -1 and sy.n_id_1(30, 30) == -1   # unsigned wrap / range guard
    assert sy.n_id_1(2, 9) == 167 and sy.n_id_1(9, 2) == 167   # subframe 5 swaps the pair
    assert sy.n_id_1(5, 20) == 0                               # quirk q5: unreachable pair reads 0


def test_fft9728_matches_numpy(oracle):
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(9728) + 1j * rng.standard_normal(9728)).astype(np.complex64)
    y = oracle.fft9728(x)
    ref = np.fft.fft(x.astype(np.complex128))
    assert np.max(np.abs(y - ref)) / np.max(np.abs(ref)) < 1e-6
    back = oracle.fft9728(y, inverse=True) / 9728
    assert np.max(np.abs(back - x)) < 1e-5


def test_find_pss_matches_direct_convolution(oracle):
    x, _, _ = load_frame("6prb")
    for r in range(3):
        q = oracle.Pss(r)
        pos, psr, pv = q.find_pss(x[:9600])
        h = oracle.pss_taps(r).astype(np.complex128)
        a = 0.2 * np.abs(np.convolve(x[:9600].astype(np.complex128), h)[:9726]) ** 2
        assert pos == int(np.argmax(a))
        assert np.allclose(q.avg()[:9726], a, rtol=2e-4, atol=1e-9)
        assert pv == pytest.approx(a[pos], rel=1e-5)
    # matched root peaks one past the PSS symbol of slot 0, PSR 4.94
    pos, psr, _ = oracle.Pss(0).find_pss(x[:9600])
    assert pos == 960 and psr == pytest.approx(4.93, abs=0.02)


def test_cfo_estimate_and_correct(oracle):
    x, _, _ = load_frame("6prb")
    q = oracle.Pss(0)
    sym = x[832:960]
    assert abs(q.cfo_compute(sym)) < 1e-3
    n = np.arange(128)
    for f in (0.1, -0.25, 0.4):
        assert q.cfo_compute((sym * np.exp(2j * np.pi * f * n / 128)).astype(np.complex64)) == pytest.approx(f, abs=5e-3)
    w = x[:9600]
    y = oracle.cfo_correct(w, 1.0 / 4096)                      # one table step per sample
    tab = oracle.cexptab()
    assert np.array_equal(y, (tab[np.arange(9600) % 4096] * w).astype(np.complex64)) or \
        np.allclose(y, tab[np.arange(9600) % 4096] * w, atol=1e-6)
    assert np.array_equal(oracle.cfo_correct(w, 0.0), w)


@pytest.mark.parametrize("name", list(FIXTURES))
def test_oracle_is_pinned_by_reference_fixture(oracle, name):
    """qa_downlink_trigger_c topology: file looped to 1 s of samples, [resampler 1/N], threshold 4.
    The chain of the cell's own root must tag exactly the advertised cell_id with normal CP, and the
    other two roots must stay silent."""
    x, decim, expect = load_frame(name)
    nsamp = {1: 1_920_000, 4: 7_680_000, 8: 15_360_000, 16: 30_720_000}[decim]
    reps = nsamp // x.size
    d = oracle.decimate(np.tile(x, reps), decim)
    assert d.size == 1_920_000
    cid = expect["cell_id"]
    for root in range(3):
        recs = oracle.chain_run(d, root, psr_threshold=4.0)
        tagged = recs[(recs["flags"] & oracle.F_TAG_CELL) != 0]
        if root == cid % 3:
            assert tagged.size > 150
            assert np.all(tagged["cell_id"] == cid)
            assert np.all((tagged["flags"] & oracle.F_CP_EXT) == 0)
            m = set(zip(tagged["m0"].tolist(), tagged["m1"].tolist()))
            m0, m1 = oracle.sss_m0m1_of(cid // 3)
            assert m == {(m0, m1), (m1, m0)}                    # subframe 0 / subframe 5
            assert tagged["call_idx"][0] <= 20                  # first tag within ~100 ms of signal
        else:
            assert tagged.size == 0
            assert float(recs["psr"].max()) < 2.0


def test_decimator_design_matches_gnuradio_shape(oracle):
    for decim, ntaps in ((4, 131), (8, 263), (16, 525)):
        t = oracle.decim_taps(decim)
        assert t.size == ntaps
        assert np.allclose(t, t[::-1])
        assert float(t.sum()) == pytest.approx(1.0, abs=1e-6)
        H = np.abs(np.fft.fft(t, 1 << 16))
        f = np.arange(1 << 16) / float(1 << 16)
        assert H[(f < 0.4 / decim)].min() > 0.975              # passband edge (fractional_bw 0.4)
        assert H[(f > 0.55 / decim) & (f < 0.5)].max() < 10 ** (-60 / 20)  # >= 60 dB anti-alias


def test_chain_end_of_stream_rules(oracle):
    x, _, _ = load_frame("6prb")
    assert len(oracle.chain_run(x[:9599], 0)) == 0
    assert len(oracle.chain_run(x[:9600], 0)) == 1
    r = oracle.chain_run(np.zeros(0, np.complex64), 0)
    assert len(r) == 0


def test_first_cell_of_a_channel_is_the_fixture_cell(oracle):
    """bench.py's `config5` leg decides "detected" with first_cell(): every chain's mib stage walks its emitted half-frames
    in call order, the first successful PBCH decode publishes, the channel's first `track` is the one earliest in the
    stream (lib/mib_impl.cc:97-183 with exit_on_success).  On the oracle's records of a bundled frame that is the frame's
    own cell -- and nothing when the threshold keeps every chain from tracking."""
    import bench
    from oracle import mib_oracle as M
    x, _, expect = load_frame("6prb")
    iq = np.tile(x, 40)

    def hf_of(recs_src):
        def hf(r):
            p0 = int(r["out_pos"])
            out = np.zeros(oracle.HALF_FRAME, np.complex64)
            lo, hi = max(p0, 0), min(p0 + oracle.HALF_FRAME, iq.size)
            out[lo - p0:hi - p0] = iq[lo:hi]
            return oracle.cfo_correct(out, float(np.float32(-r["mean_cfo"]) / np.float32(128)))
        return hf

    recs = oracle.batch_run(iq[None, :], nthreads=3)
    cell = bench.first_cell(recs[0:3], hf_of(recs), M.decode_mib, oracle)
    assert cell == (expect["cell_id"], expect["nof_prb"], 1)
    quiet = oracle.batch_run(iq[None, :], psr_threshold=1e9, nthreads=3)
    assert bench.first_cell(quiet[0:3], hf_of(quiet), M.decode_mib, oracle) is None


def test_compare_chain_records_every_tie_and_excuses_only_tiny_gaps(oracle, frames_192):
    """oracle.compare_chain (the comparison bench.py, tools/stress_parity.py and the full-size GPU tests use): records on the
    parity bar pass; a different peak position is excused only if the two lags' averages differ by at most 1e-5 relative in
    the oracle's own average at that call -- and the gap is reported either way; any other difference fails."""
    x = frames_192["6prb"]
    want = oracle.chain_run(x, 0)
    assert oracle.compare_chain(want.copy(), want, x, 0) == (True, None)
    # a metric field inside the tolerance passes, outside it fails
    got = want.copy()
    got["psr"] *= np.float32(1.00005)
    assert oracle.compare_chain(got, want, x, 0) == (True, None)
    got["psr"] *= np.float32(1.001)
    assert oracle.compare_chain(got, want, x, 0)[0] is False
    # a peak position that is NOT a near-tie: fails, with the (large) gap on record
    call = int(np.nonzero(want["flags"] & oracle.F_SEARCHED)[0][0])
    got = want.copy()
    got["peak_pos"][call] = (int(want["peak_pos"][call]) + 4000) % 9000
    ok, tie = oracle.compare_chain(got, want, x, 0)
    assert ok is False and tie["call"] == call and tie["gap"] > 1e-3 and tie["lag_oracle"] == int(want["peak_pos"][call])
    # a true twin: a lag whose average equals the peak's bit for bit is excused (gap 0) -- only up to that call is compared
    avg_gap = oracle.argmax_tie_gap(x, 0, want, call, int(want["peak_pos"][call]))
    assert avg_gap == 0.0
    got = want.copy()
    got["peak_pos"][call] = int(want["peak_pos"][call])  # same lag: no difference at all
    assert oracle.compare_chain(got, want, x, 0) == (True, None)
    # an index field other than the peak position never counts as a tie
    got = want.copy()
    got["score"][call] += 1
    assert oracle.compare_chain(got, want, x, 0) == (False, None)
    # a missing record fails
    assert oracle.compare_chain(want[:-1], want, x, 0)[0] is False
