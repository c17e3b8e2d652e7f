"""BASELINE.json configs 2 and 5 on the GPU, where the driver's `pytest -m gpu` sees them.

config 2: one 50-PRB channel at 15.36 Msps through the device decimator (x8) and the batched path: cell 125,
          50 PRB, one port, normal CP (python/qa_downlink_trigger_c.py:137-168).
config 5: the 100-PRB fixture at 30.72 Msps plus AWGN, decimated x16 on the device, threshold 4.  SNR as in
          examples/snr_ltetrigger.grc:612-663: mean signal power over mean noise power across the full sampled
          band, both taken before the sum.  Three SNR points across the knee of the detection curve
          (profiles/r03_configs_1_2_5.json: 0 % at -6 dB, 64 % at -4 dB, 100 % at -2 dB), seeded trials: the GPU path
          has to detect in exactly the trials in which the CPU oracle does, with every record field on the parity bar.
          The knee point itself (-4 dB) runs at the config's full length, 1 s per trial; the two beside it 0.25 s.
          (The whole curve -- -10..+10 dB, 50 trials, 1 s -- is bench.py's `config5` leg.)
"""
import numpy as np
import pytest

from conftest import FIXTURES, load_frame
from test_gpu_parity import assert_records_match

pytestmark = pytest.mark.gpu


def _oracle_cell(O, M, x192):
    """First cell a downlink_trigger_c built on the oracle would publish (lib/mib_impl.cc:127-177)."""
    best = None
    for root in range(3):
        recs, hfs = O.chain_run(x192, root, want_halfframes=True)
        em = recs[(recs["flags"] & O.F_EMITTED) != 0]
        for j in range(len(em)):
            if (em["flags"][j] & O.F_TAG_LOST) or not (em["flags"][j] & O.F_TAG_CELL):
                continue
            m = M.decode_mib(hfs[j][:1920], int(em["cell_id"][j]), bool(em["flags"][j] & O.F_CP_EXT))
            if m is not None:
                cand = (int(em["call_idx"][j]), dict(cell_id=int(em["cell_id"][j]), nof_prb=m["nof_prb"],
                                                     nof_tx_ports=m["nof_tx_ports"]))
                if best is None or cand[0] < best[0]:
                    best = cand
                break
    return None if best is None else best[1]


def test_config2_single_channel_50prb(lt, oracle):
    import torch
    dev = torch.device("cuda", 0)
    raw, decim, expect = load_frame("50prb")
    assert decim == 8
    xin = torch.from_numpy(np.tile(raw, 30)).to(dev)  # 0.3 s at 15.36 Msps
    y = torch.empty((1, xin.numel() // 8), dtype=torch.complex64, device=dev)
    s = torch.cuda.current_stream(dev)
    lt.Decim(8).run_device(torch.view_as_real(xin).data_ptr(), torch.view_as_real(y).data_ptr(), 1, xin.numel(), s.cuda_stream)
    bt = lt.BatchTrigger(1, y.shape[1], psr_threshold=4.0, exit_on_success=True)
    recs, events = bt.process_device(y)
    assert events[0] and events[0][0][0] == "track"
    cell = events[0][0][1]
    assert cell["cell_id"] == 125 and cell["nof_prb"] == 50 and cell["nof_tx_ports"] == 1
    assert cell["cp_len"] == "Normal" and cell["phich_len"] == "Normal" and cell["nof_phich_resources"] == "1"
    want = oracle.batch_run(y.cpu().numpy(), nthreads=3)
    for c in range(3):
        assert_records_match(recs[c], want[c], f"chain {c}")


def test_config5_snr_sweep_detection_equals_oracle(lt, oracle):
    import torch
    from oracle import mib_oracle as M
    dev = torch.device("cuda", 0)
    raw, decim, _ = load_frame("100prb")
    assert decim == 16
    frames_of = {-6: 25, -4: 100, -2: 25}  # 0.25 s per trial (a 20 dB cell is published after ~0.1 s); the knee point 1 s
    sig_full = torch.from_numpy(np.tile(raw, max(frames_of.values()))).to(dev)
    p_sig = float(np.mean(np.abs(raw) ** 2))  # over the full 30.72 MHz band
    dec = lt.Decim(16)
    trials = 10
    s = torch.cuda.current_stream(dev)
    rates = {}
    # One noise realisation per trial, drawn on the host (numpy's generator gives the same samples on every box; the
    # device generator's stream depends on the GPU it runs on) and scaled per SNR: the records compared below are the
    # same on every run.
    noises = []
    for t in range(trials):
        rng = np.random.default_rng(500000 + t)
        nz = rng.standard_normal((sig_full.numel(), 2), dtype=np.float32)
        noises.append(torch.view_as_complex(torch.from_numpy(nz).to(dev)))
    for snr in (-6, -4, -2):
        sig = sig_full[:raw.size * frames_of[snr]]
        n_in, n_out = sig.numel(), sig.numel() // 16
        bt = lt.BatchTrigger(trials, n_out, psr_threshold=4.0, exit_on_success=True)
        y = torch.empty((trials, n_out), dtype=torch.complex64, device=dev)
        for t in range(trials):
            xin = sig + noises[t][:n_in] * float(np.sqrt(p_sig / (10.0 ** (snr / 10.0)) / 2.0))
            dec.run_device(torch.view_as_real(xin).data_ptr(), torch.view_as_real(y[t]).data_ptr(), 1, n_in, s.cuda_stream)
        torch.cuda.synchronize()
        recs, events = bt.process_device(y)
        gpu_cell = [ev[0][1] if ev else None for ev in events]
        yh = y.cpu().numpy()
        want = oracle.batch_run(yh, nthreads=8)
        for c in range(3 * trials):
            assert_records_match(recs[c], want[c], f"SNR {snr} dB trial {c // 3} root {c % 3}")
        det = []
        for t in range(trials):
            oc = _oracle_cell(oracle, M, yh[t])
            g = gpu_cell[t]
            assert (g is None) == (oc is None), f"SNR {snr} dB trial {t}: GPU {g} vs oracle {oc}"
            if g is not None:
                assert (g["cell_id"], g["nof_prb"], g["nof_tx_ports"]) == (oc["cell_id"], oc["nof_prb"], oc["nof_tx_ports"])
            det.append(bool(g and g["cell_id"] == 369 and g["nof_prb"] == 100))
        rates[snr] = sum(det) / trials
        if snr == -4:
            # the knee point once more at window-cache level 4 (lag-metric streams: what bench.py's config5 leg runs at) --
            # twice, the second pass comes with the second run --: the oracle's records, the same detections
            bt.batch.set_window_cache(4)
            for _ in range(2):
                recs4, events4 = bt.process_device(y)
            assert bt.batch.cache_stats()[1] > 100, "searches out of the streams"
            for c in range(3 * trials):
                assert_records_match(recs4[c], want[c], f"level 4, SNR {snr} dB trial {c // 3} root {c % 3}")
            assert [ev[0][1] if ev else None for ev in events4] == gpu_cell
    # the knee: detection does not get worse with SNR (between the two points of one length; the knee point has four times
    # the time to find its cell), and the top point detects (mostly)
    assert rates[-6] <= rates[-2] + 0.2 and rates[-6] <= rates[-4] + 0.2, rates
    assert rates[-2] >= 0.5 and rates[-4] >= 0.3, rates
