"""The per-call C-ABI under GNU Radio's threading model (SURVEY 8(b) "Threading").

The reference runs every block on a thread of its own -- three pss, three sss and three mib blocks per trigger
(python/downlink_trigger_c.py:27-45) -- and polls the pss getters from yet another thread without a lock
(lib/pss_impl.h:95-100).  The contract of include/ltetrigger_hip.h: distinct handles may be used from distinct threads
at once; lt_last_error() is per thread; the plain getters may be read while their handle is busy.

Nine threads (ctypes releases the GIL around every call), each with its own lt_pss / lt_cfo / lt_sync handles, run
find_pss -> cfo_compute -> cfo_correct -> detect_cp -> m0m1 -> N_id_1 over 200 half-frames of the four fixtures at once,
while a tenth thread polls getters of their handles and provokes errors of its own.  Every thread's results must be
the serial run's, bit for bit; no thread sees another's error text.

And one level up (SURVEY 8(e) "one host thread per GPU"): one process, one lt_batch per device, one host thread each --
skipped on a box with a single GPU.
"""
import threading

import numpy as np
import pytest

from conftest import FIXTURES
import siggen

pytestmark = pytest.mark.gpu

N_HALF_FRAMES = 200


def _block_calls(lt, x, n_id_2, handles=None, tick=None):
    """What one pss block + one sss block do per half-frame, through the srsLTE-shaped calls (lib/pss_impl.cc:165-204,
    lib/sss_impl.cc:104-124), on consecutive 9600-sample windows of x."""
    pss, cfo, sync = handles if handles else (lt.Pss(n_id_2), lt.Cfo(), lt.Sync(n_id_2))
    out = []
    for k in range(N_HALF_FRAMES):
        w = x[k * 9600:(k + 1) * 9600]
        pos, psr, pv = pss.find_pss(w)
        sym = np.zeros(128, np.complex64)
        lo = max(pos - 128, 0)
        sym[128 - (pos - lo):] = w[lo:pos]
        c = pss.cfo_compute(sym)
        y = cfo.correct(w, float(np.float32(-c) / np.float32(128)))
        cp = sync.detect_cp(y, 960)
        sync.set_cp(cp)
        idx = 960 - 256 - sync.cp_len()
        m0, v0, m1, v1 = sync.sss_m0m1(y[idx:idx + 128])
        nid1 = sync.n_id_1(m0, m1)
        out.append((pos, np.float32(psr).tobytes(), np.float32(pv).tobytes(), np.float32(c).tobytes(), y.tobytes(), cp,
                    m0, np.float32(v0).tobytes(), m1, np.float32(v1).tobytes(), nid1, sync.cell_id(), sync.cp_metrics()))
        if tick is not None:
            tick[0] += 1
    return out


def test_nine_blocks_on_nine_threads_and_a_poller(lt, frames_192):
    import ctypes as C
    names = list(FIXTURES)
    rng = np.random.default_rng(7)
    # (fixture, root) per thread: every fixture with its own root and two others; steady-state frame looped, a little noise
    jobs = []
    for i in range(9):
        name = names[i % 4]
        frame = frames_192[name][19200:38400]
        x = np.tile(frame, N_HALF_FRAMES // 2)
        x = (x + 0.05 * (rng.standard_normal(x.size) + 1j * rng.standard_normal(x.size))).astype(np.complex64)
        jobs.append((x, (FIXTURES[name][2]["cell_id"] + i // 4) % 3))
    serial = [_block_calls(lt, x, r) for x, r in jobs]
    # (this thread's lt_last_error() may hold the text of an earlier test's expected failure: the workers below are new
    # threads, whose text starts out empty)

    handles = [(lt.Pss(r), lt.Cfo(), lt.Sync(r)) for _, r in jobs]
    results, errors, ticks = [None] * 9, [None] * 9, [[0] for _ in range(9)]
    start = threading.Barrier(10)
    done = threading.Event()
    polled = {"n": 0, "own_error_seen": 0, "foreign": 0}

    def worker(i):
        try:
            start.wait()
            results[i] = _block_calls(lt, jobs[i][0], jobs[i][1], handles[i], ticks[i])
            errors[i] = lt.last_error()
        except Exception as e:  # noqa: BLE001
            errors[i] = f"exception: {e!r}"

    def poller():
        L = lt.lib()
        start.wait()
        pv = C.c_float()
        while not done.is_set():
            for pss, _, sync in handles:
                L.lt_pss_get_peak_value(pss._h, C.byref(pv))      # getters of handles that are busy on other threads
                L.lt_sync_get_cp_len(sync._h)
                L.lt_sync_get_cell_id(sync._h)
            rc = L.lt_pss_set_N_id_2(handles[0][0]._h, 7)          # an error of this thread's own
            msg = lt.last_error()
            if rc == -2 and "invalid N_id_2 7" in msg:
                polled["own_error_seen"] += 1
            else:
                polled["foreign"] += 1
            polled["n"] += 1

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(9)] + [threading.Thread(target=poller)]
    for t in threads:
        t.start()
    for t in threads[:9]:
        t.join(timeout=600)
    done.set()
    threads[9].join(timeout=60)
    assert all(not t.is_alive() for t in threads)
    assert polled["n"] > 0 and polled["foreign"] == 0 and polled["own_error_seen"] == polled["n"], polled
    for i in range(9):
        assert errors[i] == "", f"thread {i}: lt_last_error() = {errors[i]!r} (another thread's text leaked, or a call failed)"
        assert ticks[i][0] == N_HALF_FRAMES
        assert len(results[i]) == len(serial[i])
        for k, (a, b) in enumerate(zip(results[i], serial[i])):
            assert a == b, f"thread {i} half-frame {k}: concurrent run differs from the serial one"
    # the workers really overlapped in time is not asserted (scheduling); that they ran on nine handles at once is the
    # point: no hidden globals, one stream per handle, hipSetDevice per call


def test_one_process_one_batch_per_device(lt, oracle, frames_192):
    """one host thread per GPU, no collective.  Every device runs the same four channels; every device's
    records are the single-device run's, bit for bit."""
    n_dev = lt.lib().lt_device_count()
    if n_dev < 2:
        pytest.skip(f"needs two GPUs in one process (lt_device_count() = {n_dev})")
    base = [(frames_192[n][19200:38400], FIXTURES[n][2]["cell_id"]) for n in FIXTURES]
    iq, _ = siggen.scan_channels(base, 4, 40 * 9600 + 77, snr_db=10.0, cfo=True, seed0=910)
    ref_b = lt.Batch(4, iq.shape[1], device=0)
    ref_b.run_host(iq)
    ref = [r.copy() for r in ref_b.collect()]
    out, errs = [None] * n_dev, [None] * n_dev

    def run(d):
        try:
            b = lt.Batch(4, iq.shape[1], device=d)
            for _ in range(3):
                b.reset()
                b.run_host(iq)
            out[d] = [r.copy() for r in b.collect()]
            errs[d] = ""
        except Exception as e:  # noqa: BLE001
            errs[d] = repr(e)

    ts = [threading.Thread(target=run, args=(d,)) for d in range(n_dev)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=600)
    for d in range(n_dev):
        assert errs[d] == "", f"device {d}: {errs[d]}"
        for c in range(12):
            assert out[d][c].tobytes() == ref[c].tobytes(), f"device {d} chain {c}"
