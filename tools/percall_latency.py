#!/usr/bin/env python3
"""Latency of the per-call (srsLTE-shaped) entry points the drop-in blocks use: one half-frame per call,
host buffers in and out (what a GNU Radio work() call pays) -- from one thread, and from nine threads at once, each with
handles of its own (a downlink_trigger_c under GNU Radio's thread-per-block scheduler: three pss, three sss blocks busy
at the same time, getters polled beside them)."""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ltetrigger_amd as lt  # noqa: E402


def calls(x):
    p, s, c = lt.Pss(0), lt.Sync(0), lt.Cfo()
    return (("lt_pss_find_pss", lambda: p.find_pss(x)),
            ("lt_pss_cfo_compute", lambda: p.cfo_compute(x[832:960])),
            ("lt_cfo_correct (9600)", lambda: c.correct(x, 1e-4)),
            ("lt_sync_detect_cp", lambda: s.detect_cp(x, 960)),
            ("lt_sss_m0m1_partial", lambda: s.sss_m0m1(x[695:823])))


def main():
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(9600) + 1j * rng.standard_normal(9600)).astype(np.complex64)
    one = calls(x)
    for _ in range(20):
        one[0][1]()
    n = 300
    print("one thread:")
    for name, fn in one:
        fn()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        print(f"  {name:24s} {(time.perf_counter() - t0) / n * 1e6:8.1f} us per call")
    # nine threads, every one with its own handles (and so its own stream), the same sequence of calls per half-frame
    sets = [calls(x) for _ in range(9)]
    for cs in sets:
        for _, fn in cs:
            fn()
    out = [None] * 9
    go = threading.Barrier(9)

    def work(i):
        go.wait()
        t = {}
        for name, fn in sets[i]:
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            t[name] = (time.perf_counter() - t0) / n
        out[i] = t
    th = [threading.Thread(target=work, args=(i,)) for i in range(9)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    wall = time.perf_counter() - t0
    print("nine threads at once (own handles each; ctypes releases the GIL inside a call):")
    for name, _ in one:
        v = [o[name] for o in out]
        print(f"  {name:24s} {np.mean(v) * 1e6:8.1f} us per call (slowest thread {max(v) * 1e6:.1f})")
    print(f"  9 x 5 x {n} calls in {wall * 1e3:.0f} ms: {9 * 5 * n / wall / 1e3:.1f} k calls/s")


if __name__ == "__main__":
    main()
