#!/usr/bin/env python3
"""The decimator alone: lt_decim_kernel on one channel of 15.36 M samples (rate 8) or 30.72 M (rate 16), timed with HIP
events; a few launches, for rocprofv3 (tools/pmc_scan.sh python3 lt_decim tools/decim_probe.py 8)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ltetrigger_amd as lt  # noqa: E402

rate = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n_ch = int(sys.argv[2]) if len(sys.argv) > 2 else 1
n_in = 1920000 * rate
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
x = torch.view_as_complex(torch.randn((n_ch, n_in, 2), device=dev, generator=g))
y = torch.empty((n_ch, n_in // rate), dtype=torch.complex64, device=dev)
dec = lt.Decim(rate)
s = torch.cuda.Stream(dev)
torch.cuda.synchronize()
times = []
with torch.cuda.stream(s):
    for _ in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        dec.run_device(torch.view_as_real(x).data_ptr(), torch.view_as_real(y).data_ptr(), n_ch, n_in, s.cuda_stream)
        e1.record(s)
        s.synchronize()
        times.append(e0.elapsed_time(e1))
ms = min(times)
print(f"decimation {rate}, {n_ch} channel(s) x {n_in} samples: {ms * 1e3:.1f} us per launch "
      f"({n_ch * n_in * 8 / ms / 1e6:.0f} GB/s of input, {n_ch * n_in * (8 + 8 / rate) / ms / 1e6:.0f} GB/s read + written "
      f"[the algorithmic bytes, 8 + 8/D per input sample], {n_ch * n_in / ms / 1e6:.1f} Gsamp/s)")
