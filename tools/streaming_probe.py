#!/usr/bin/env python3
"""bench.py's streaming leg alone, with the slowest chunks of the timed pass listed (index, microseconds)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import ltetrigger_amd as lt  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    frames = bench.base_frames(lt)
    n_ch, n = 64, 1920000
    iq, _ = bench.make_channels(torch, frames, n_ch, n, 0, 20.0, dev)
    chunk = 19200
    n_chunks = n // chunk
    host = torch.empty((n_chunks, n_ch, chunk), dtype=torch.complex64, pin_memory=True)
    host.copy_(iq.reshape(n_ch, n_chunks, chunk).permute(1, 0, 2))
    torch.cuda.synchronize()
    dbuf = [torch.empty((n_ch, chunk), dtype=torch.complex64, device=dev) for _ in range(2)]
    b = lt.Batch(n_ch, chunk, psr_threshold=4.0, device=0)
    b.set_phase_timing(0)
    s_run, s_copy = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    copied = [torch.cuda.Event() for _ in range(2)]
    freed = [torch.cuda.Event() for _ in range(2)]
    for rep in range(3):
        b.reset()
        t = np.zeros((n_chunks, 4))
        with torch.cuda.stream(s_copy):
            dbuf[0].copy_(host[0], non_blocking=True)
            copied[0].record(s_copy)
        for k in range(n_chunks):
            t0 = time.perf_counter()
            if k + 1 < n_chunks:
                with torch.cuda.stream(s_copy):
                    if k + 1 >= 2:
                        s_copy.wait_event(freed[(k + 1) & 1])
                    dbuf[(k + 1) & 1].copy_(host[k + 1], non_blocking=True)
                    copied[(k + 1) & 1].record(s_copy)
            t1 = time.perf_counter()
            s_run.wait_event(copied[k & 1])
            b.push_device(torch.view_as_real(dbuf[k & 1]).data_ptr(), chunk, end_of_stream=(k == n_chunks - 1), stream=s_run.cuda_stream)
            freed[k & 1].record(s_run)
            t2 = time.perf_counter()
            b.collect()
            t3 = time.perf_counter()
            t[k] = (t1 - t0, t2 - t1, t3 - t2, t3 - t0)
        us = t * 1e6
        worst = np.argsort(-us[:, 3])[:5]
        print(f"pass {rep}: mean {us[:, 3].mean():.1f} us, median {np.median(us[:, 3]):.1f}; slowest chunks (index: feed / push / collect / all): "
              + "; ".join(f"{int(i)}: {us[i, 0]:.0f} / {us[i, 1]:.0f} / {us[i, 2]:.0f} / {us[i, 3]:.0f}" for i in worst), flush=True)
    b.close()


if __name__ == "__main__":
    main()
