#!/usr/bin/env python3
"""Kernel-time probe for lt_chain_kernel: pure search (noise only) vs the bench mix."""
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
    ap.add_argument("--mode", default="noise", choices=["noise", "clean", "mix", "decim"])
    ap.add_argument("--ablate", type=int, default=0,
                    help="bit mask. Diagnostics build of the chain kernel (fused mode unless 128 is set): 1 no window / slice load, "
                         "2 no matched filter (fused), 4 no reductions (fused), 8 in-kernel stamps, "
                         "16 direct-form filter (fused), 32 no ring mean, 64 no SSS stage, 128 replay kernel instead of fused. "
                         "Product kernels: 256 per-chain shader clocks, 512 / 1024 lag-metric kernel without stores / loads")
    ap.add_argument("--threshold", type=float, default=4.0)
    ap.add_argument("--fused", action="store_true", help="fused per-chain kernel instead of bulk + replay")
    args = ap.parse_args()
    import torch
    import ltetrigger_amd as lt
    import bench
    dev = torch.device("cuda", 0)
    n = int(args.seconds * 1_920_000)
    if args.mode == "decim":
        # traffic calibration: the decimator reads n_in*8 B with 8 B/lane coalesced loads (the access
        # shape of the chain kernel's window loads) and writes n_in*8/D B
        g = torch.Generator(device=dev); g.manual_seed(1)
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
        g = torch.Generator(device=dev); g.manual_seed(1)
        iq = torch.view_as_complex(torch.randn((args.channels, n, 2), generator=g, device=dev) * 0.5)
    elif args.mode == "clean":
        iq, _ = bench.make_channels(torch, frames, args.channels, n, 0, 100.0, dev)
        for c in range(7, args.channels, 8):  # make every channel carry a cell
            iq[c] = iq[c - 1]
    else:
        iq, _ = bench.make_channels(torch, frames, args.channels, n, 0, 20.0, dev)
    iqr = torch.view_as_real(iq).contiguous()
    torch.cuda.synchronize()
    b = lt.Batch(args.channels, n, psr_threshold=args.threshold)
    if args.fused:
        b.set_mode(0)
    if args.ablate:
        import ctypes
        lt.lib().lt_debug_set_ablation(b._h, ctypes.c_int(args.ablate))
    s = torch.cuda.Stream(device=dev)
    ms = []
    for _ in range(args.reps):
        b.reset()
        b.run_device(iqr.data_ptr(), n, s.cuda_stream)
        ms.append(b.last_kernel_ms())
    recs = b.collect()
    nsearch = sum(int(((r["flags"] & lt.F_SEARCHED) != 0).sum()) for r in recs)
    nemit_tr = sum(int((((r["flags"] & lt.F_SSS_RAN) != 0)).sum()) for r in recs)
    per_chain_search = max(int(((r["flags"] & lt.F_SEARCHED) != 0).sum()) for r in recs)
    k = float(np.median(ms))
    if args.ablate & 256:
        import ctypes
        nchains = args.channels * 3
        prof = (ctypes.c_ulonglong * (nchains * 24))()
        lt.lib().lt_debug_get_profile(b._h, prof)
        el = np.array(prof[:], dtype=np.float64).reshape(nchains, 24)[:, 23] / 1e3
        kinds = []
        for c in range(nchains):
            r = recs[c]
            tr = int(((r['flags'] & lt.F_SSS_RAN) != 0).sum())
            kinds.append('tracked' if tr > 20 else 'search')
        kinds = np.array(kinds)
        for kd in ('tracked', 'search'):
            v = el[kinds == kd]
            if v.size:
                print(f'{kd} chains ({v.size}): elapsed kclk min {v.min():.0f} median {np.median(v):.0f} max {v.max():.0f}')
    if (args.ablate & 8) and (args.ablate & 0xFF):
        import ctypes
        nchains = args.channels * 3
        prof = (ctypes.c_ulonglong * (nchains * 24))()
        lt.lib().lt_debug_get_profile(b._h, prof)
        p = np.array(prof[:], dtype=np.float64).reshape(nchains, 24)
        names = ["conv", "ema+mirror", "barrier", "argmax", "walk", "side", "load"]
        tnames = ["stage", "cfo", "mean", "rotate", "sss"]
        busiest = int(np.argmax(p[:, :7].sum(1) + p[:, 8:13].sum(1)))
        tracker = int(np.argmax(p[:, 13]))
        for c in {busiest, tracker, 0}:
            nw, ne = max(p[c, 7], 1), max(p[c, 13], 1)
            print(f"chain {c}: total {(p[c, :7].sum() + p[c, 8:13].sum()) / 1e3:.0f} kcycles; per searched window ({int(p[c, 7])}): " +
                  ", ".join(f"{n}={p[c, i] / nw:.0f}" for i, n in enumerate(names)) +
                  f"; per tracked emit ({int(p[c, 13])}): " +
                  ", ".join(f"{n}={p[c, 8 + i] / ne:.0f}" for i, n in enumerate(tnames)) +
                  f" | wave0: cfo={p[c, 9] / ne:.0f} ringmean={p[c, 21] / ne:.0f} (slow path {int(p[c, 22])}x) | sss: cp={p[c, 17] / ne:.0f} dft={p[c, 18] / ne:.0f} bar={p[c, 19] / ne:.0f} m0m1={p[c, 20] / ne:.0f}" +
                  f" | batches: {int(p[c, 17])} x {p[c, 16] / max(p[c, 17], 1):.0f} clk" +
                  " | conv: " + ", ".join(f"{n}={p[c, 14 + i] / nw:.0f}" for i, n in
                  enumerate(["inputs+barrier", "transforms", "metric+barriers"])))
    print(f"ablate={args.ablate} mode={args.mode} channels={args.channels} n={n}: kernel median {k:.3f} ms (min {min(ms):.3f}); "
          f"searches total {nsearch} (max/chain {per_chain_search}), tracked emits {nemit_tr}; "
          f"{args.channels * n / k / 1e6:.1f} Gsamp/s; {k * 1e3 / max(per_chain_search, 1):.2f} us per window of the busiest chain")


if __name__ == "__main__":
    main()
