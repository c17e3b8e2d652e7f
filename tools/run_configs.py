#!/usr/bin/env python3
"""BASELINE.json configs 1, 2 and 5 end to end (config 3 / 4 are bench.py).

  config 1  cell_search_file contract on lte_frame_6prb_cellid_123 @1.92 Msps, CPU oracle (plumbing)
  config 2  single channel, 50 PRB fixture @15.36 Msps looped to 1 s: GPU decimator x8 -> lt_batch_run
            -> host MIB; verify cell_id = 125, nof_prb = 50
  config 5  100 PRB fixture @30.72 Msps + AWGN, SNR -10..+10 dB, threshold 4, N seeded trials per
            point: detection-rate-vs-SNR for the GPU path and for the CPU oracle, trial for trial

Writes one JSON document to stdout (and to --out).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden", "test_frames")


def oracle_cell(O, M, x192, root_hint=None):
    """First cell a downlink_trigger_c built from the oracle would publish: (cell dict | None)."""
    best = None
    for root in range(3):
        recs, hfs = O.chain_run(x192, root, want_halfframes=True)
        em = recs[(recs["flags"] & O.F_EMITTED) != 0]
        for j in range(len(em)):
            if em["flags"][j] & O.F_TAG_LOST:
                continue
            if not (em["flags"][j] & O.F_TAG_CELL):
                continue
            m = M.decode_mib(hfs[j][:1920], int(em["cell_id"][j]), bool(em["flags"][j] & O.F_CP_EXT))
            if m is not None:
                cand = (int(em["call_idx"][j]), dict(cell_id=int(em["cell_id"][j]), nof_prb=m["nof_prb"],
                                                     nof_tx_ports=m["nof_tx_ports"], cp_len="Normal"))
                if best is None or cand[0] < best[0]:
                    best = cand
                break
    return None if best is None else best[1]


def _load_stress():
    import importlib.util
    spec = importlib.util.spec_from_file_location("stress_parity", os.path.join(os.path.dirname(os.path.abspath(__file__)), "stress_parity.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def main():
    stress = _load_stress()
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=50)
    ap.add_argument("--out", default=None)
    ap.add_argument("--skip-oracle", action="store_true")
    ap.add_argument("--snrs", default="", help="config 5: comma-separated SNRs in dB (default -10..10 in steps of 2)")
    args = ap.parse_args()
    import torch
    import ltetrigger_amd as lt
    from oracle import oracle as O
    from oracle import mib_oracle as M
    dev = torch.device("cuda", 0)
    res = {}

    # ---------------- config 1 (CPU oracle, plumbing)
    x = np.fromfile(os.path.join(GOLD, "lte_frame_6prb_cellid_123"), dtype=np.complex64)
    t0 = time.perf_counter()
    cell = oracle_cell(O, M, np.tile(x, 100))
    res["config1"] = {"input": "lte_frame_6prb_cellid_123 looped x100 @1.92 Msps", "path": "CPU oracle",
                      "cell": cell, "seconds": round(time.perf_counter() - t0, 3),
                      "ok": bool(cell and cell["cell_id"] == 123 and cell["nof_prb"] == 6)}

    # ---------------- config 2 (single 15.36 Msps channel on the GPU)
    raw = np.fromfile(os.path.join(GOLD, "lte_frame_50prb_cellid_125"), dtype=np.complex64)
    xin = torch.from_numpy(np.tile(raw, 100)).to(dev)  # 15.36 M samples, resident in HBM
    y = torch.empty((1, xin.numel() // 8), dtype=torch.complex64, device=dev)
    dec = lt.Decim(8)
    bt = lt.BatchTrigger(1, y.shape[1], psr_threshold=4.0, exit_on_success=True)
    s = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    times = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        dec.run_device(torch.view_as_real(xin).data_ptr(), torch.view_as_real(y).data_ptr(), 1, xin.numel(),
                       s.cuda_stream)
        bt.batch.reset()
        bt.batch.run_device(torch.view_as_real(y).data_ptr(), y.shape[1], s.cuda_stream)
        e1.record(s)
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1))
    recs, events = bt.process_device(y, s.cuda_stream)
    cell = events[0][0][1] if events[0] else None
    ms = float(np.median(times))
    res["config2"] = {"input": "lte_frame_50prb_cellid_125 looped x100 @15.36 Msps (15.36 M samples)",
                      "path": "lt_decim_kernel(x8) -> lt_batch_run (scan rounds, 3 chains) -> host MIB",
                      "cell": cell, "device_ms": round(ms, 3),
                      "msamp_per_s_at_input_rate": round(xin.numel() / ms / 1e3, 1),
                      "ok": bool(cell and cell["cell_id"] == 125 and cell["nof_prb"] == 50
                                 and cell["nof_tx_ports"] == 1 and cell["cp_len"] == "Normal")}
    del xin, y

    # ---------------- config 5 (SNR sweep at 30.72 Msps)
    raw = np.fromfile(os.path.join(GOLD, "lte_frame_100prb_cellid_369"), dtype=np.complex64)
    sig = torch.from_numpy(np.tile(raw, 100)).to(dev)  # 30.72 M samples = 1 s
    p_sig = float(np.mean(np.abs(raw) ** 2))
    n_in = sig.numel()
    n_out = n_in // 16
    dec = lt.Decim(16)
    T = args.trials
    bt = lt.BatchTrigger(T, n_out, psr_threshold=4.0, exit_on_success=True)
    snrs = [int(v) for v in args.snrs.split(",")] if args.snrs else list(range(-10, 11, 2))
    curve = []
    for snr in snrs:
        y = torch.empty((T, n_out), dtype=torch.complex64, device=dev)
        for t in range(T):
            g = torch.Generator(device=dev)
            g.manual_seed(100000 + 1000 * (snr + 10) + t)
            noise = torch.view_as_complex(torch.randn((n_in, 2), generator=g, device=dev))
            xin = sig + noise * float(np.sqrt(p_sig / (10.0 ** (snr / 10.0)) / 2.0))
            dec.run_device(torch.view_as_real(xin).data_ptr(), torch.view_as_real(y[t]).data_ptr(), 1, n_in, None)
            torch.cuda.synchronize()
        recs, events = bt.process_device(y)
        gpu_det = [bool(ev and ev[0][0] == "track" and ev[0][1]["cell_id"] == 369 and ev[0][1]["nof_prb"] == 100)
                   for ev in events]
        gpu_wrong = sum(1 for ev in events if ev and not (ev[0][1]["cell_id"] == 369 and ev[0][1]["nof_prb"] == 100))
        point = {"snr_db": snr, "trials": T, "gpu_detected": int(sum(gpu_det)), "gpu_wrong_cell": gpu_wrong,
                 "gpu_rate": round(sum(gpu_det) / T, 3)}
        if not args.skip_oracle:
            yh = y.cpu().numpy()
            want = O.batch_run(yh, nthreads=min(os.cpu_count() or 1, 16))
            same_records = True
            differing = []
            cpu_det = []
            for t in range(T):
                ok = False
                for root in range(3):
                    g_, w_ = recs[t * 3 + root], want[t * 3 + root]
                    d = stress.first_difference(g_, w_)
                    if d:
                        item = {"trial": t, "root": root, "what": d}
                        if d.startswith("peak_pos at call "):
                            call = int(d.split()[3].rstrip(":"))
                            item["near_tie"] = bool(stress.near_tie(O, yh[t], root, w_, call, g_["peak_pos"][call]))
                        # an arg-max between two lags whose averages agree to 1e-4 is inside the lag metric's tolerance
                        if not item.get("near_tie"):
                            same_records = False
                            if args.out and len(differing) < 3:
                                np.save(os.path.splitext(args.out)[0] + f"_snr{snr}_trial{t}.npy", yh[t])
                        differing.append(item)
                c = oracle_cell(O, M, yh[t]) if gpu_det[t] or True else None
                ok = bool(c and c["cell_id"] == 369 and c["nof_prb"] == 100)
                cpu_det.append(ok)
            point.update({"cpu_detected": int(sum(cpu_det)), "cpu_rate": round(sum(cpu_det) / T, 3),
                          "trial_for_trial_equal": bool(cpu_det == gpu_det), "records_equal": same_records,
                          "record_differences": differing})
        curve.append(point)
        print(json.dumps(point), file=sys.stderr, flush=True)
    res["config5"] = {"input": "lte_frame_100prb_cellid_369 looped x100 @30.72 Msps + AWGN (SNR over the full band), "
                               "GPU decimator x16, threshold 4, 1 s per trial",
                      "detection": "first track dict has cell_id 369 and nof_prb 100", "curve": curve}
    out = json.dumps(res, indent=1)
    print(out)
    if args.out:
        with open(args.out, "w") as f:
            f.write(out)


if __name__ == "__main__":
    main()
