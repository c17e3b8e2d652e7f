#!/usr/bin/env python3
"""Small batches: the scan rounds from the window cache (lt_metric_kernel + lt_scan_cached_kernel) against the scan rounds through
the transforms (lt_scan_kernel<G>), N channels x 1 s at 20 dB, device time of one lt_batch_run (median of 8, HIP events).
LT_CACHED_MAXCH in the environment sets the batch size up to which the cached form is used (experiment)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import ltetrigger_amd as lt  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    frames = bench.base_frames(lt)
    n = 1920000
    for n_ch in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 6, 8, 12, 16, 24, 32]:
        iq, _ = bench.make_channels(torch, frames, n_ch, n, 0, 20.0, dev)
        b = lt.Batch(n_ch, n, psr_threshold=4.0, device=0)
        s = torch.cuda.Stream(device=dev)
        torch.cuda.synchronize()
        t = []
        for _ in range(10):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            b.reset()
            e0.record(s)
            b.run_device(torch.view_as_real(iq).data_ptr(), n, s.cuda_stream)
            e1.record(s)
            torch.cuda.synchronize()
            t.append(e0.elapsed_time(e1))
        recs = b.collect()
        h = 0
        for r in recs:
            h = (h * 1000003 + int(np.frombuffer(r.tobytes(), np.uint8).astype(np.uint64).sum())) % (1 << 61)
        print(f"{n_ch:3d} channels: {1e3 * float(np.median(t[2:])):8.1f} us, {b.last_launch_count()} launches, records checksum {h:x}", flush=True)
        b.close()


if __name__ == "__main__":
    main()
