#!/usr/bin/env python3
"""Stage-time probe of the batched path: noise only, clean cells, the bench mix, or the decimator alone
(the calibration load of tools/pmc_traffic.sh).  Prints the lt_batch_phase_ms stages and the wall time per run."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--channels", type=int, default=64)
    ap.add_argument("--seconds", type=float, default=1.0)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--mode", default="mix", choices=["noise", "clean", "mix", "decim"])
    ap.add_argument("--threshold", type=float, default=4.0)
    ap.add_argument("--snr-db", type=float, default=20.0, help="mode mix: SNR of the channels that carry a cell")
    ap.add_argument("--exec-mode", type=int, default=1, help="lt_batch_set_mode: 1 = grid scan + per-chain kernel, 0 = per-chain kernel only")
    ap.add_argument("--scan-segments", default="", help="first,later windows per scan round")
    args = ap.parse_args()
    import torch
    import ltetrigger_amd as lt
    import bench
    dev = torch.device("cuda", 0)
    n = int(args.seconds * 1_920_000)
    if args.mode == "decim":
        # traffic calibration: the decimator reads n_in*8 B with 8 B/lane coalesced loads (the access
        # shape of the scan kernel's block loads) and writes n_in*8/D B
        g = torch.Generator(device=dev)
        g.manual_seed(1)
        x = torch.randn((args.channels, n, 2), generator=g, device=dev)
        y = torch.empty((args.channels, n // 8, 2), device=dev)
        d = lt.Decim(8)
        s = torch.cuda.Stream(device=dev)
        torch.cuda.synchronize()
        for _ in range(args.reps):
            d.run_device(x.data_ptr(), y.data_ptr(), args.channels, n, s.cuda_stream)
        torch.cuda.synchronize()
        print(f"mode=decim channels={args.channels} n={n}: reads {args.channels * n * 8} B, writes {args.channels * (n // 8) * 8} B per launch")
        return
    frames = bench.base_frames(lt)
    if args.mode == "noise":
        g = torch.Generator(device=dev)
        g.manual_seed(1)
        iq = torch.view_as_complex(torch.randn((args.channels, n, 2), generator=g, device=dev) * 0.5)
    elif args.mode == "clean":
        iq, _ = bench.make_channels(torch, frames, args.channels, n, 0, 100.0, dev)
        for c in range(7, args.channels, 8):  # make every channel carry a cell
            iq[c] = iq[c - 1]
    else:
        iq, _ = bench.make_channels(torch, frames, args.channels, n, 0, args.snr_db, dev)
    iqr = torch.view_as_real(iq).contiguous()
    torch.cuda.synchronize()
    b = lt.Batch(args.channels, n, psr_threshold=args.threshold)
    b.set_mode(args.exec_mode)
    if args.scan_segments:
        b.set_scan_segments(*[int(v) for v in args.scan_segments.split(",")])
    s = torch.cuda.Stream(device=dev)
    ms, ph = [], []
    for _ in range(args.reps):
        b.reset()
        b.run_device(iqr.data_ptr(), n, s.cuda_stream)
        torch.cuda.synchronize()
        ms.append(b.last_kernel_ms())
        ph.append(b.phase_ms(0))
    recs = b.collect()
    calls = sum(len(r) for r in recs)
    searched = sum(int(((r["flags"] & lt.F_SEARCHED) != 0).sum()) for r in recs)
    tagged = sum(int(((r["flags"] & lt.F_TAG_CELL) != 0).sum()) for r in recs)
    k = float(np.median(ms))
    p = np.median(np.array(ph), axis=0)
    print(f"mode={args.mode} exec_mode={args.exec_mode} channels={args.channels} n={n}: {k:.3f} ms per run (min {min(ms):.3f}); "
          f"scan rounds {p[0]:.3f}, per-chain kernel {p[2]:.3f}, per-record kernels {p[3]:.3f} ms; "
          f"{args.channels * n / k / 1e3:.0f} Msamp/s; {calls} calls, {searched} searches, {tagged} cell tags")


if __name__ == "__main__":
    main()
