#!/usr/bin/env python3
"""BASELINE configs[1] (one 50-PRB channel at 15.36 Msps, 1 s) a few times over, for `rocprofv3 --kernel-trace`:
tools/timeline.py then prints the kernels of the last pass.  Prints the median device time of a pass (HIP events)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import ltetrigger_amd as lt  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    raw = np.fromfile(os.path.join(ROOT, "tests", "golden", "test_frames", "lte_frame_50prb_cellid_125"), dtype=np.complex64)
    xin = torch.from_numpy(np.tile(raw, 100)).to(dev)
    n_in, n_out = xin.numel(), xin.numel() // 8
    y = torch.empty((1, n_out), dtype=torch.complex64, device=dev)
    dec = lt.Decim(8, device=0)
    b = lt.Batch(1, n_out, psr_threshold=4.0, device=0)
    if len(sys.argv) > 1:
        b.set_window_cache(int(sys.argv[1]))
    s = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    t = []
    for _ in range(8):
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        b.reset()
        e0.record(s)
        dec.run_device(torch.view_as_real(xin).data_ptr(), torch.view_as_real(y).data_ptr(), 1, n_in, s.cuda_stream)
        e1.record(s)
        b.run_device(torch.view_as_real(y).data_ptr(), n_out, s.cuda_stream)
        e2.record(s)
        torch.cuda.synchronize()
        t.append((e0.elapsed_time(e1), e1.elapsed_time(e2)))
    recs = b.collect()
    cells = set()
    for r in recs:
        cells |= set(r["cell_id"][(r["flags"] & lt.F_TAG_CELL) != 0].tolist())
    print("decimator %.1f us, path %.1f us, launches %d, cells %s" % (1e3 * float(np.median([a for a, _ in t[2:]])), 1e3 * float(np.median([p for _, p in t[2:]])), b.last_launch_count(), sorted(int(c) for c in cells)))


if __name__ == "__main__":
    main()
