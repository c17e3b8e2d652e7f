"""PBCH/MIB stage: the product's C++ decoder (host/pbch.cc, via ltr_decode_mib)
against the numpy oracle, both pinned by the MIB fields the reference's tests assert on the four
fixtures, and both exercised on synthetic frames (1/2/4 ports, both CP types, every bandwidth)."""
import numpy as np
import pytest

from conftest import FIXTURES, load_frame
import lte_synth

KEYS = ("nof_prb", "nof_tx_ports", "phich_len", "nof_phich_resources", "sfn_offset", "frame_idx")


def _aligned_halfframes(oracle, d, cid):
    recs, hfs = oracle.chain_run(d, cid % 3, want_halfframes=True)
    em = recs[(recs["flags"] & oracle.F_EMITTED) != 0]
    idx = np.nonzero((em["flags"] & oracle.F_TAG_CELL) != 0)[0]
    return em, hfs, idx


@pytest.mark.parametrize("name", list(FIXTURES))
def test_fixture_mib_is_pinned(oracle, lt, name):
    """cell dict fields of python/qa_downlink_trigger_c.py:46-65,93-98 (...203)."""
    from oracle import mib_oracle
    x, decim, expect = load_frame(name)
    d = oracle.decimate(np.tile(x, 24), decim)
    em, hfs, idx = _aligned_halfframes(oracle, d, expect["cell_id"])
    got_o, got_c, misses = None, None, 0
    for j in idx[:4]:
        sf = hfs[j][:1920]
        cid = int(em["cell_id"][j])
        o = mib_oracle.decode_mib(sf, cid)
        c = lt.decode_mib(sf, cid)
        assert (o is None) == (c is None)          # subframe 5 carries no PBCH
        if o is None:
            misses += 1
            continue
        assert {k: o[k] for k in KEYS} == {k: c[k] for k in KEYS}
        got_o, got_c = o, c
    assert got_c is not None and misses >= 1
    assert got_c["nof_prb"] == expect["nof_prb"]
    assert got_c["nof_tx_ports"] == 1
    assert got_c["phich_len"] == "Normal"
    assert got_c["nof_phich_resources"] == "1"
    assert got_c["sfn_offset"] == 0                # README.rst:97-104
    # SURVEY Appendix B.3 MIB bits
    bw = {6: "000", 25: "010", 50: "011", 100: "101"}[expect["nof_prb"]]
    assert got_o["bits"] == bw + "010" + "0" * 18


CASES = [
    # cell_id, cp_ext, ports, nof_prb, phich_ext, phich_r, sfn
    (0, False, 1, 6, 0, 2, 0),
    (503, False, 2, 100, 1, 0, 8),
    (77, False, 4, 15, 0, 3, 1022),
    (301, True, 1, 75, 0, 1, 3),
    (257, True, 4, 25, 1, 2, 517),
    (168, True, 2, 50, 0, 0, 6),
]


@pytest.mark.parametrize("cid,ext,ports,prb,pext,pr,sfn", CASES)
def test_synthetic_mib_round_trip(oracle, lt, cid, ext, ports, prb, pext, pr, sfn):
    from oracle import mib_oracle
    ch = np.array([1.0, 0.7 * np.exp(1.1j), 0.5 * np.exp(-2.0j), 0.9 * np.exp(0.3j)])[:ports]
    f = lte_synth.make_frame(cid, cp_ext=ext, nof_ports=ports, nof_prb=prb, phich_ext=pext, phich_r=pr,
                             sfn=sfn, channel=ch, seed=cid)
    want = dict(nof_prb=prb, nof_tx_ports=ports, phich_len="Extended" if pext else "Normal",
                nof_phich_resources=["1/6", "1/2", "1", "2"][pr], sfn_offset=(sfn >> 2) << 2,
                frame_idx=sfn % 4)
    # the file starts on the frame boundary: subframe 0 is the first 1920 samples
    c = lt.decode_mib(f[:1920], cid, ext)
    assert c == want
    o = mib_oracle.decode_mib(f[:1920], cid, ext)
    assert {k: o[k] for k in KEYS} == want
    # with a port-count hint only that count is tried (srslte_ue_mib_set_cell semantics)
    assert lt.decode_mib(f[:1920], cid, ext, ports) == want
    wrong = {1: 2, 2: 4, 4: 1}[ports]
    assert lt.decode_mib(f[:1920], cid, ext, wrong) is None
    # wrong cell id / wrong CP / subframe 5: nothing decodes
    assert lt.decode_mib(f[:1920], (cid + 1) % 504, ext) is None
    assert lt.decode_mib(f[9600:9600 + 1920], cid, ext) is None


def test_mib_noise_only_never_decodes(lt):
    rng = np.random.default_rng(0)
    for k in range(8):
        n = (rng.standard_normal(1920) + 1j * rng.standard_normal(1920)).astype(np.complex64)
        assert lt.decode_mib(n, 3 * k + 1, bool(k & 1)) is None
    assert lt.decode_mib(np.zeros(1920, np.complex64), 0) is None
