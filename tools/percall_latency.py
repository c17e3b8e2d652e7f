#!/usr/bin/env python3
"""Latency of the per-call (srsLTE-shaped) entry points the drop-in blocks use: one half-frame per call,
host buffers in and out (what a GNU Radio work() call pays)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ltetrigger_amd as lt  # noqa: E402


def main():
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(9600) + 1j * rng.standard_normal(9600)).astype(np.complex64)
    p = lt.Pss(0)
    s = lt.Sync(0)
    c = lt.Cfo()
    for _ in range(20):
        p.find_pss(x)
    for name, fn, n in (("lt_pss_find_pss", lambda: p.find_pss(x), 300),
                        ("lt_pss_cfo_compute", lambda: p.cfo_compute(x[832:960]), 300),
                        ("lt_cfo_correct (9600)", lambda: c.correct(x, 1e-4), 300),
                        ("lt_sync_detect_cp", lambda: s.detect_cp(x, 960), 300),
                        ("lt_sss_m0m1_partial", lambda: s.sss_m0m1(x[695:823]), 300)):
        fn()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        dt = (time.perf_counter() - t0) / n
        print(f"{name:24s} {dt * 1e6:8.1f} us per call")


if __name__ == "__main__":
    main()
