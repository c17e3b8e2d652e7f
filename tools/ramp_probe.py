#!/usr/bin/env python3
"""Why are a cold process's first ~25 steps 7 % slower than the settled ones?  Per-step device time (HIP events) of 60
consecutive steps of the bench workload, (a) straight after set-up, (b) straight after 80 ms of unrelated GPU work (the
decimator on buffers of its own): if (b) starts settled, it is the clock the chip ramps up; if (b) shows the same ramp, it is
something the path's own buffers go through."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import ltetrigger_amd as lt  # noqa: E402


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "a"
    dev = torch.device("cuda:0")
    frames = bench.base_frames(lt)
    n_ch, n = 64, 1920000
    b = lt.Batch(n_ch, n, psr_threshold=4.0, device=0)
    iq, _ = bench.make_channels(torch, frames, n_ch, n, 0, 20.0, dev)
    s = torch.cuda.Stream(device=dev)
    if mode == "b":
        x = torch.view_as_complex(torch.randn((8, 15360000, 2), device=dev))
        y = torch.empty((8, 1920000), dtype=torch.complex64, device=dev)
        dec = lt.Decim(8)
        torch.cuda.synchronize()
        for _ in range(240):  # ~ 80 ms of a loaded queue
            dec.run_device(torch.view_as_real(x).data_ptr(), torch.view_as_real(y).data_ptr(), 8, 15360000, s.cuda_stream)
    elif mode == "c":  # the same path on a batch and a capture of its own: the same kernels, none of this batch's buffers
        b2 = lt.Batch(n_ch, n, psr_threshold=4.0, device=0)
        iq2, _ = bench.make_channels(torch, frames, n_ch, n, 64, 20.0, dev)
        torch.cuda.synchronize()
        for _ in range(100):
            b2.reset()
            b2.run_device(torch.view_as_real(iq2).data_ptr(), n, s.cuda_stream)
    elif mode == "d":  # this batch's own steps, then a pause of 50 ms with the GPU idle
        for _ in range(100):
            b.reset()
            b.run_device(torch.view_as_real(iq).data_ptr(), n, s.cuda_stream)
        torch.cuda.synchronize()
        import time
        time.sleep(0.05)
    else:
        torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(61)]
    ev[0].record(s)
    for k in range(60):
        b.reset()
        b.run_device(torch.view_as_real(iq).data_ptr(), n, s.cuda_stream)
        ev[k + 1].record(s)
    torch.cuda.synchronize()
    t = np.array([ev[k].elapsed_time(ev[k + 1]) for k in range(60)]) * 1e3
    print(f"mode {mode}: steps 0-4 {t[:5].mean():.0f} us, 5-24 {t[5:25].mean():.0f}, 25-39 {t[25:40].mean():.0f}, 40-59 {t[40:].mean():.0f}; first ten: " + " ".join(f"{v:.0f}" for v in t[:10]))


if __name__ == "__main__":
    main()
