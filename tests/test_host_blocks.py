"""The gr::ltetrigger host blocks on the in-repo runtime, driven from Python exactly like the
reference's own tests drive the real blocks (python/qa_downlink_trigger_c.py:67-203), plus
block-level parity of the pss / sss output streams and tags against the oracle."""
import json
import os

import numpy as np
import pytest

from conftest import FIXTURES, GOLDEN, load_frame
import siggen

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", list(FIXTURES))
def test_qa_downlink_trigger_c(lt, name):
    """file_source(repeat) -> head(1 s) -> [rational_resampler_ccc(1, N)] -> downlink_trigger_c(4, True);
    message_debug on 'track'; the same assertions as the reference's test_*prb_t."""
    fname, decim, expect = FIXTURES[name]
    srate = 1.92e6 * decim
    tb = lt.TopBlock()
    fsrc = tb.file_source(os.path.join(GOLDEN, "test_frames", fname), repeat=True)
    head = tb.head(int(srate))  # 1 sec of data
    dltrig = lt.downlink_trigger_c(tb, 4, True)
    if decim == 1:
        tb.connect(fsrc, head, dltrig)
    else:
        resampler = tb.rational_resampler_ccc(1, decim)
        tb.connect(fsrc, head, resampler, dltrig)
    msgdebug = tb.message_debug()
    tb.msg_connect(dltrig, "track", msgdebug, "store")
    tb.run()
    nmsgs = msgdebug.num_messages()
    assert nmsgs, "No message triggered"
    cell = msgdebug.get_message(0)
    assert cell["cell_id"] == expect["cell_id"], "wrong cell_id"
    assert cell["cp_len"] == "Normal", "wrong cp_len"
    assert cell["nof_phich_resources"] == "1", "wrong nof_phich_resources"
    assert cell["nof_prb"] == expect["nof_prb"], "wrong nof_prb"
    assert cell["nof_tx_ports"] == 1, "wrong nof_tx_ports"
    assert cell["phich_len"] == "Normal", "wrong phich_len"
    assert cell["sfn_offset"] == 0
    assert set(cell) == {"cell_id", "nof_tx_ports", "cp_len", "nof_prb", "phich_len",
                         "nof_phich_resources", "sfn_offset", "tracking_start_time"}
    # getters of the chain that found the cell (include/ltetrigger/pss.h:72-87)
    p = dltrig.pss[expect["cell_id"] % 3]
    assert p.max_psr() > 4 and p.mean_psr() > 4 and p.tracking_score() == 16.0
    assert abs(p.mean_cfo()) < 0.05 and p.psr_threshold() == 4.0
    for k in range(3):
        if k != expect["cell_id"] % 3:
            assert dltrig.pss[k].max_psr() < 2.5 and dltrig.pss[k].tracking_score() == 0.0


def test_threshold_clamp(lt):
    tb = lt.TopBlock()
    t = lt.downlink_trigger_c(tb, 0.3)
    assert t.psr_threshold == lt.MIN_PSR_THRESHOLD == 1.5
    assert all(p.psr_threshold() == 1.5 for p in t.pss)
    t.set_psr_threshold(7.0)
    assert all(p.psr_threshold() == 7.0 for p in t.pss)
    t.set_psr_threshold(1.0)
    assert all(p.psr_threshold() == 1.5 for p in t.pss)


def _oracle_view(oracle, d, root, **kw):
    recs, hfs = oracle.chain_run(d, root, want_halfframes=True, **kw)
    em = recs[(recs["flags"] & oracle.F_EMITTED) != 0]
    return recs, em, hfs


@pytest.mark.parametrize("case", ["clean", "cfo_noise", "dropout"])
def test_pss_and_sss_block_streams_match_oracle(lt, oracle, case):
    """pss block output = concatenated oracle half-frames, bit for bit, with "tracking_lost" tags at
    the same offsets; sss passes the stream through and adds cell_id / cp_type tags at the same
    half-frames as the oracle's records."""
    x, decim, expect = load_frame("25prb")
    d = oracle.decimate(np.tile(x, 40), decim)
    rng = np.random.default_rng(11)
    if case == "cfo_noise":
        d = siggen.apply_cfo(d, -0.11) + siggen.awgn(rng, d.size, 0.5 / 10 ** 1.5)
    elif case == "dropout":
        d = d.copy()
        d[22 * 9600:42 * 9600] = siggen.awgn(rng, 20 * 9600, 0.5)
    root = expect["cell_id"] % 3
    tb = lt.TopBlock()
    src = tb.vector_source(d)
    pss = tb.pss(root, 4.0)
    sss = tb.sss(root)
    p1 = tb.probe()
    p2 = tb.probe()
    tb.connect(src, pss, p1)
    tb.connect(pss, sss, p2)
    tb.run()
    recs, em, hfs = _oracle_view(oracle, d, root)
    got = p1.items()
    assert got.size == hfs.size
    assert np.array_equal(got.view(np.uint32), hfs.reshape(-1).view(np.uint32))
    assert np.array_equal(p2.items().view(np.uint32), got.view(np.uint32))   # sss is a pass-through
    lost = sorted(t["offset"] for t in p1.tags() if t["key"] == "tracking_lost")
    want_lost = [i * 9600 for i in range(len(em)) if em["flags"][i] & oracle.F_TAG_LOST]
    assert lost == want_lost
    tags2 = p2.tags()
    cell = {t["offset"]: t["value"] for t in tags2 if t["key"] == "cell_id"}
    cp = {t["offset"]: t["value"] for t in tags2 if t["key"] == "cp_type"}
    want_cell = {i * 9600: int(em["cell_id"][i]) for i in range(len(em)) if em["flags"][i] & oracle.F_TAG_CELL}
    assert cell == want_cell
    assert set(cp) == set(cell) and all(v is True for v in cp.values())
    assert sorted(t["offset"] for t in tags2 if t["key"] == "tracking_lost") == want_lost  # TPP_ALL_TO_ALL
    # getters against the oracle's last record
    assert pss.max_psr() == pytest.approx(float(recs["psr_max"][-1]), rel=1e-4)
    assert pss.tracking_score() == float(recs["score"][-1])
    if case == "dropout":
        assert want_lost and any(o > 22 * 9600 for o in want_lost)


def test_track_and_drop_messages_and_cellstore(lt, oracle):
    """A cell appears, disappears (drop) and comes back (second track); cellstore follows."""
    x, _, _ = load_frame("6prb")
    rng = np.random.default_rng(5)
    # while tracking the block searches only every 9th half-frame and the EMA carries the old peak,
    # so it takes a long outage (here 70 half-frames) before the chain declares the cell lost
    sig = siggen.loop_to(x, 170 * 9600).copy()
    sig[30 * 9600:100 * 9600] = siggen.awgn(rng, 70 * 9600, 0.5)
    tb = lt.TopBlock()
    src = tb.vector_source(sig)
    trig = lt.downlink_trigger_c(tb, 4, exit_on_success=False)
    store = tb.cellstore()
    tracks, drops = tb.message_debug(), tb.message_debug()
    tb.connect(src, trig)
    tb.msg_connect(trig, "track", store, "track")
    tb.msg_connect(trig, "drop", store, "drop")
    tb.msg_connect(trig, "track", tracks, "store")
    tb.msg_connect(trig, "drop", drops, "store")
    tb.run()
    assert tracks.num_messages() == 2 and drops.num_messages() == 1
    assert tracks.get_message(0)["cell_id"] == 123 and drops.get_message(0)["cell_id"] == 123
    assert store.tracking() and len(store.cells()) == 1
    assert store.latest_cell()["nof_prb"] == 6


def test_batch_trigger_equals_block_path(lt, oracle):
    """BatchTrigger (device state machine + host MIB replay) reports the same cells at the same
    general_work call as the block-by-block flowgraph."""
    rows = []
    for name in FIXTURES:
        x, decim, _ = load_frame(name)
        rows.append(oracle.decimate(np.tile(x, 30), decim))
    iq = np.stack(rows)
    bt = lt.BatchTrigger(4, iq.shape[1])
    recs, events = bt.process(iq)
    for ch, name in enumerate(FIXTURES):
        expect = FIXTURES[name][2]
        ev = events[ch]
        assert [e[0] for e in ev] == ["track"]
        cell = ev[0][1]
        assert cell["cell_id"] == expect["cell_id"] and cell["nof_prb"] == expect["nof_prb"]
        assert cell["nof_tx_ports"] == 1 and cell["cp_len"] == "Normal"
        tb = lt.TopBlock()
        src = tb.vector_source(iq[ch])
        trig = lt.downlink_trigger_c(tb, 4, exit_on_success=False)
        md = tb.message_debug()
        tb.connect(src, trig)
        tb.msg_connect(trig, "track", md, "store")
        tb.run()
        assert md.num_messages() == 1
        m = md.get_message(0)
        for k in ("cell_id", "nof_tx_ports", "cp_len", "nof_prb", "phich_len", "nof_phich_resources", "sfn_offset"):
            assert m[k] == cell[k]


def test_cell_search_file_cli(lt, capsys, tmp_path):
    """examples/cell_search_file.py contract: 'Starting cell search... done.' + JSON with status."""
    import importlib.util
    path = os.path.join(os.path.dirname(lt.__file__), "examples", "cell_search_file.py")
    spec = importlib.util.spec_from_file_location("cell_search_file", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    f = os.path.join(GOLDEN, "test_frames", "lte_frame_50prb_cellid_125")
    res = mod.main(mod.parse_args([f, "-s", "15.36M", "--repeat", "-c", "1.92M"]))
    out = capsys.readouterr().out
    assert out.startswith("Starting cell search... done.")
    assert len(res) == 1
    cell = json.loads(res[0])
    assert cell["status"] == "FOUND" and cell["cell_id"] == 125 and cell["nof_prb"] == 50
    assert cell["nof_tx_ports"] == 1 and cell["cp_len"] == "Normal" and cell["sfn_offset"] == 0
    # noise only -> NOT_FOUND
    noise = tmp_path / "noise.cf32"
    rng = np.random.default_rng(0)
    (rng.standard_normal(2 * 192000).astype(np.float32) * 0.5).tofile(noise)
    res = mod.main(mod.parse_args([str(noise), "-s", "1.92M"]))
    assert [json.loads(r) for r in res] == [{"status": "NOT_FOUND"}]
    with pytest.raises(SystemExit):
        mod.main(mod.parse_args([f, "-s", "2M"]))


def test_thread_per_block_scheduler_gives_the_same_messages_and_streams(lt, oracle):
    """GNU Radio runs the reference one thread per block, and applications poll the pss getters and the cellstore from
    their own thread meanwhile (lib/pss_impl.h:95-100; examples/rtlsdr_ltetrigger.grc:751-838).  The same flowgraphs under
    the in-repo runtime's thread-per-block scheduler (top_block::run_threaded: twelve block threads per trigger, each pss /
    sss with its own C-ABI handles and streams) while a Python thread polls: the pss / sss output streams and tags, the
    track / drop messages and the cellstore are those of the single-threaded run."""
    import threading
    x, _, _ = load_frame("6prb")
    rng = np.random.default_rng(5)
    sig = siggen.loop_to(x, 170 * 9600).copy()
    sig[30 * 9600:100 * 9600] = siggen.awgn(rng, 70 * 9600, 0.5)

    def run(threaded):
        tb = lt.TopBlock()
        src = tb.vector_source(sig)
        trig = lt.downlink_trigger_c(tb, 4, exit_on_success=False)
        store = tb.cellstore()
        tracks, drops = tb.message_debug(), tb.message_debug()
        tb.connect(src, trig)
        tb.msg_connect(trig, "track", store, "track")
        tb.msg_connect(trig, "drop", store, "drop")
        tb.msg_connect(trig, "track", tracks, "store")
        tb.msg_connect(trig, "drop", drops, "store")
        # a second consumer of the same source: one chain's streams, tapped behind pss and behind sss
        pss, sss = tb.pss(0, 4.0), tb.sss(0)
        p1, p2 = tb.probe(), tb.probe()
        tb.connect(src, pss, p1)
        tb.connect(pss, sss, p2)
        stop, polls = threading.Event(), [0]

        def poll():
            while not stop.is_set():
                for p in trig.pss + [pss]:
                    p.max_psr(), p.mean_psr(), p.mean_cfo(), p.tracking_score(), p.psr_threshold()
                store.tracking(), store.cells(), store.latest_cell(), tracks.num_messages()
                polls[0] += 1
        t = threading.Thread(target=poll)
        if threaded:
            t.start()
        tb.run(threaded=threaded)
        stop.set()
        if threaded:
            t.join()
        strip = lambda m: {k: v for k, v in m.items() if k != "tracking_start_time"}
        return {"tracks": [strip(tracks.get_message(i)) for i in range(tracks.num_messages())],
                "drops": [strip(drops.get_message(i)) for i in range(drops.num_messages())],
                "cells": [strip(c) for c in store.cells()], "pss": p1.items(), "sss": p2.items(),
                "tags1": sorted((t["offset"], t["key"]) for t in p1.tags()),
                "tags2": sorted((t["offset"], t["key"], str(t["value"])) for t in p2.tags()),
                "getters": (pss.max_psr(), pss.mean_psr(), pss.mean_cfo(), pss.tracking_score()), "polls": polls[0]}
    ref = run(False)
    assert len(ref["tracks"]) == 2 and len(ref["drops"]) == 1 and ref["pss"].size > 50 * 9600
    for rep in range(2):
        got = run(True)
        assert got["polls"] > 0
        for k in ("tracks", "drops", "cells", "tags1", "tags2", "getters"):
            assert got[k] == ref[k], f"threaded run {rep}: {k}"
        for k in ("pss", "sss"):
            assert np.array_equal(got[k].view(np.uint32), ref[k].view(np.uint32)), f"threaded run {rep}: {k} stream"
