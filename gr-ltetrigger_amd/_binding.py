"""ctypes binding of include/ltetrigger_hip.h (libltetrigger_hip.so, gfx950 kernels).

There is no CPU implementation behind these classes: if the shared library is
missing, or no gfx950 device is usable, they raise.
"""
import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (LTETRIGGER_HIP_LIB: another build of the same library, for A/B timing on one box -- tools/ab.sh; never a CPU stand-in)
LIB_PATH = os.environ.get("LTETRIGGER_HIP_LIB") or os.path.join(_HERE, "libltetrigger_hip.so")

LT_OK, LT_ERROR, LT_ERROR_INVALID_INPUTS = 0, -1, -2
HALF_FRAME, SLOT, SYMBOL_SZ, CONV_LEN, NLAGS = 9600, 960, 128, 9728, 9726
CP_NORM, CP_EXT = 0, 1

F_SEARCHED, F_OVER, F_TRACKING, F_EMITTED = 1, 2, 4, 8
F_TAG_LOST, F_TAG_CELL, F_CP_EXT, F_SSS_RAN = 16, 32, 64, 128

# lt_record
RECORD_DTYPE = np.dtype(
    [
        ("in_pos", "<i8"), ("out_pos", "<i8"),
        ("channel", "<i4"), ("n_id_2", "<i4"), ("call_idx", "<i4"), ("flags", "<u4"),
        ("peak_pos", "<i4"), ("score", "<i4"),
        ("psr", "<f4"), ("peak_value", "<f4"), ("cfo", "<f4"), ("mean_cfo", "<f4"),
        ("m_norm_avg", "<f4"), ("m_ext_avg", "<f4"), ("m0_value", "<f4"), ("m1_value", "<f4"),
        ("m0", "<i4"), ("m1", "<i4"), ("n_id_1", "<i4"), ("cell_id", "<i4"),
        ("psr_max", "<f4"), ("timer", "<i4"),
    ]
)

# every symbol include/ltetrigger_hip.h declares
EXPORTS = [
    "lt_version", "lt_last_error", "lt_device_count",
    "lt_pss_init", "lt_pss_set_N_id_2", "lt_pss_free", "lt_pss_reset", "lt_pss_find_pss",
    "lt_pss_get_peak_value", "lt_pss_get_avg", "lt_pss_cfo_compute",
    "lt_cfo_init", "lt_cfo_free", "lt_cfo_set_last_freq", "lt_cfo_correct",
    "lt_sync_init", "lt_sync_set_N_id_2", "lt_sync_free", "lt_sync_reset", "lt_sync_detect_cp",
    "lt_sync_set_cp", "lt_sync_get_cp_len", "lt_sync_get_cp_metrics", "lt_sss_m0m1_partial",
    "lt_sss_N_id_1", "lt_sync_get_cell_id",
    "lt_batch_create", "lt_batch_free", "lt_batch_reset", "lt_batch_set_mode", "lt_batch_set_psr_threshold",
    "lt_batch_run", "lt_batch_collect", "lt_batch_collect_previous", "lt_batch_record_capacity", "lt_batch_enable_halfframes",
    "lt_batch_get_halfframes", "lt_batch_last_kernel_ms", "lt_batch_last_launch_count", "lt_batch_phase_ms", "lt_batch_push",
    "lt_batch_get_mode", "lt_batch_set_scan_segments", "lt_batch_set_overlap", "lt_batch_debug_hold_records", "lt_batch_set_phase_timing", "lt_batch_set_window_cache", "lt_batch_cache_stats", "lt_batch_set_scan_groups",
    "lt_decim_create", "lt_decim_free", "lt_decim_ntaps", "lt_decim_get_taps", "lt_decim_run",
]


class ChainParams(C.Structure):
    _fields_ = [("psr_threshold", C.c_float), ("track_after", C.c_int32),
                ("track_every", C.c_int32)]


class LtError(RuntimeError):
    pass


_lib = None


def lib():
    """Load libltetrigger_hip.so (built by __graft_entry__.build() / make). No fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LtError(
            f"{LIB_PATH} is missing: build it with `make -C {_HERE}` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    # PyTorch ships a HIP runtime of its own.  The two coexist in one process only in this order: PyTorch's
    # libraries loaded and its device context up first, then this library (observed on ROCm 7.2 / torch 2.10: any
    # other interleaving leaves one of the two without a device).  So if the process uses torch, bring it up now.
    if "torch" in sys.modules:
        try:
            _torch = sys.modules["torch"]
            if _torch.cuda.is_available():
                _torch.cuda.init()
        except Exception:  # torch without a usable GPU: nothing to order against
            pass
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, f32, u32 = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_uint32
    pp = C.POINTER(vp)
    L.lt_version.restype = C.c_char_p
    L.lt_last_error.restype = C.c_char_p
    L.lt_device_count.restype = i32
    L.lt_pss_init.argtypes = [pp, i32, i32, i32]
    L.lt_pss_set_N_id_2.argtypes = [vp, i32]
    L.lt_pss_free.argtypes = [vp]
    L.lt_pss_free.restype = None
    L.lt_pss_reset.argtypes = [vp]
    L.lt_pss_find_pss.argtypes = [vp, vp, C.POINTER(f32)]
    L.lt_pss_get_peak_value.argtypes = [vp, C.POINTER(f32)]
    L.lt_pss_get_avg.argtypes = [vp, vp]
    L.lt_pss_cfo_compute.argtypes = [vp, vp, C.POINTER(f32)]
    L.lt_cfo_init.argtypes = [pp, i32, i32]
    L.lt_cfo_free.argtypes = [vp]
    L.lt_cfo_free.restype = None
    L.lt_cfo_set_last_freq.argtypes = [vp, f32]
    L.lt_cfo_correct.argtypes = [vp, vp, vp, f32]
    L.lt_sync_init.argtypes = [pp, i32, i32, i32, i32]
    L.lt_sync_set_N_id_2.argtypes = [vp, i32]
    L.lt_sync_free.argtypes = [vp]
    L.lt_sync_free.restype = None
    L.lt_sync_reset.argtypes = [vp]
    L.lt_sync_detect_cp.argtypes = [vp, vp, i32]
    L.lt_sync_set_cp.argtypes = [vp, i32]
    L.lt_sync_get_cp_len.argtypes = [vp]
    L.lt_sync_get_cp_metrics.argtypes = [vp, C.POINTER(f32), C.POINTER(f32)]
    L.lt_sss_m0m1_partial.argtypes = [vp, vp, C.POINTER(u32), C.POINTER(f32), C.POINTER(u32),
                                      C.POINTER(f32)]
    L.lt_sss_N_id_1.argtypes = [vp, u32, u32]
    L.lt_sync_get_cell_id.argtypes = [vp]
    L.lt_batch_create.argtypes = [pp, i32, i32, i64, C.POINTER(ChainParams)]
    L.lt_batch_free.argtypes = [vp]
    L.lt_batch_free.restype = None
    L.lt_batch_reset.argtypes = [vp]
    L.lt_batch_set_mode.argtypes = [vp, i32]
    L.lt_batch_set_psr_threshold.argtypes = [vp, f32]
    L.lt_batch_run.argtypes = [vp, vp, i32, i64, vp]
    L.lt_batch_push.argtypes = [vp, vp, i32, i64, i32, vp]
    L.lt_batch_get_mode.argtypes = [vp]
    L.lt_batch_set_scan_segments.argtypes = [vp, i32, i32]
    L.lt_batch_set_overlap.argtypes = [vp, i32]
    L.lt_batch_debug_hold_records.argtypes = [vp, i32]
    L.lt_batch_set_phase_timing.argtypes = [vp, i32]
    L.lt_batch_set_window_cache.argtypes = [vp, i32]
    L.lt_batch_cache_stats.argtypes = [vp, vp]
    L.lt_batch_set_scan_groups.argtypes = [vp, i32]
    L.lt_batch_collect.argtypes = [vp, vp, i64, vp]
    L.lt_batch_collect_previous.argtypes = [vp, vp, i64, vp]
    L.lt_batch_record_capacity.argtypes = [i64]
    L.lt_batch_record_capacity.restype = i64
    L.lt_batch_enable_halfframes.argtypes = [vp, i64]
    L.lt_batch_get_halfframes.argtypes = [vp, i32, vp, i64, C.POINTER(i64)]
    L.lt_batch_last_kernel_ms.argtypes = [vp, C.POINTER(f32)]
    L.lt_batch_last_launch_count.argtypes = [vp, C.POINTER(i32)]
    L.lt_batch_phase_ms.argtypes = [vp, i32, C.POINTER(f32)]
    L.lt_decim_create.argtypes = [pp, i32, i32]
    L.lt_decim_free.argtypes = [vp]
    L.lt_decim_free.restype = None
    L.lt_decim_ntaps.argtypes = [vp]
    L.lt_decim_get_taps.argtypes = [vp, vp]
    L.lt_decim_run.argtypes = [vp, vp, vp, i32, i64, i32, vp]
    _lib = L
    return L


def last_error():
    return lib().lt_last_error().decode(errors="replace")


def _check(rc, what):
    if rc < 0:
        raise LtError(f"{what}: {last_error()} (rc={rc})")
    return rc


def _c64(a, n=None):
    a = np.ascontiguousarray(a, dtype=np.complex64)
    if n is not None and a.size != n:
        raise ValueError(f"expected {n} samples, got {a.size}")
    return a


class Pss:
    """srslte_pss_t replacement (call sites lib/pss_impl.cc:72-78,125,142,165,200)."""

    def __init__(self, n_id_2=None, device=0):
        h = C.c_void_p()
        self._h = None
        _check(lib().lt_pss_init(C.byref(h), device, HALF_FRAME, SYMBOL_SZ), "Error initializing PSS")
        self._h = h
        if n_id_2 is not None:
            self.set_N_id_2(n_id_2)

    def close(self):
        if self._h and lib is not None:  # (module globals are None while the interpreter shuts down)
            lib().lt_pss_free(self._h)
            self._h = None

    __del__ = close

    def set_N_id_2(self, n_id_2):
        _check(lib().lt_pss_set_N_id_2(self._h, n_id_2), "Error initializing PSS N_id_2")

    def reset(self):
        _check(lib().lt_pss_reset(self._h), "lt_pss_reset")

    def find_pss(self, x9600):
        x = _c64(x9600, HALF_FRAME)
        psr = C.c_float()
        pos = _check(lib().lt_pss_find_pss(self._h, x.ctypes.data_as(C.c_void_p), C.byref(psr)),
                     "lt_pss_find_pss")
        pv = C.c_float()
        _check(lib().lt_pss_get_peak_value(self._h, C.byref(pv)), "lt_pss_get_peak_value")
        return pos, psr.value, pv.value

    def avg(self):
        a = np.zeros(CONV_LEN + 1, np.float32)
        _check(lib().lt_pss_get_avg(self._h, a.ctypes.data_as(C.c_void_p)), "lt_pss_get_avg")
        return a

    def cfo_compute(self, sym128):
        x = _c64(sym128, SYMBOL_SZ)
        cfo = C.c_float()
        _check(lib().lt_pss_cfo_compute(self._h, x.ctypes.data_as(C.c_void_p), C.byref(cfo)),
               "lt_pss_cfo_compute")
        return cfo.value


class Cfo:
    """srslte_cfo_t replacement (lib/pss_impl.cc:78,148,204)."""

    def __init__(self, nsamples=HALF_FRAME, device=0):
        h = C.c_void_p()
        self._h = None
        _check(lib().lt_cfo_init(C.byref(h), device, nsamples), "Error initializing CFO")
        self._h = h
        self.nsamples = nsamples

    def close(self):
        if self._h and lib is not None:  # (module globals are None while the interpreter shuts down)
            lib().lt_cfo_free(self._h)
            self._h = None

    __del__ = close

    def set_last_freq(self, f):
        _check(lib().lt_cfo_set_last_freq(self._h, f), "lt_cfo_set_last_freq")

    def correct(self, x, freq):
        x = _c64(x, self.nsamples)
        y = np.empty_like(x)
        _check(lib().lt_cfo_correct(self._h, x.ctypes.data_as(C.c_void_p),
                                    y.ctypes.data_as(C.c_void_p), freq), "lt_cfo_correct")
        return y


class Sync:
    """srslte_sync_t replacement (lib/sss_impl.cc:63-69,94,104-124)."""

    def __init__(self, n_id_2=None, device=0):
        h = C.c_void_p()
        self._h = None
        _check(lib().lt_sync_init(C.byref(h), device, HALF_FRAME, 100, SYMBOL_SZ),
               "Error initializing SSS SYNC")
        self._h = h
        if n_id_2 is not None:
            self.set_N_id_2(n_id_2)

    def close(self):
        if self._h and lib is not None:  # (module globals are None while the interpreter shuts down)
            lib().lt_sync_free(self._h)
            self._h = None

    __del__ = close

    def set_N_id_2(self, n_id_2):
        _check(lib().lt_sync_set_N_id_2(self._h, n_id_2), "Error initializing SSS SYNC N_id_2")

    def reset(self):
        _check(lib().lt_sync_reset(self._h), "lt_sync_reset")

    def detect_cp(self, hf, peak_pos=SLOT):
        x = _c64(hf)
        if x.size < peak_pos:
            raise ValueError("need at least peak_pos samples")
        return _check(lib().lt_sync_detect_cp(self._h, x.ctypes.data_as(C.c_void_p), peak_pos),
                      "lt_sync_detect_cp")

    def set_cp(self, cp):
        _check(lib().lt_sync_set_cp(self._h, cp), "lt_sync_set_cp")

    def cp_len(self):
        return _check(lib().lt_sync_get_cp_len(self._h), "lt_sync_get_cp_len")

    def cp_metrics(self):
        a, b = C.c_float(), C.c_float()
        _check(lib().lt_sync_get_cp_metrics(self._h, C.byref(a), C.byref(b)), "lt_sync_get_cp_metrics")
        return a.value, b.value

    def sss_m0m1(self, sym128):
        x = _c64(sym128, SYMBOL_SZ)
        m0, m1, v0, v1 = C.c_uint32(), C.c_uint32(), C.c_float(), C.c_float()
        _check(lib().lt_sss_m0m1_partial(self._h, x.ctypes.data_as(C.c_void_p), C.byref(m0),
                                         C.byref(v0), C.byref(m1), C.byref(v1)),
               "lt_sss_m0m1_partial")
        return m0.value, v0.value, m1.value, v1.value

    def n_id_1(self, m0, m1):
        return lib().lt_sss_N_id_1(self._h, m0, m1)

    def cell_id(self):
        return lib().lt_sync_get_cell_id(self._h)


def record_capacity(n_samples):
    return int(lib().lt_batch_record_capacity(int(n_samples)))


class Batch:
    """Batched pss->sss chains: n_channels x 3 roots (grid scan of the idle chains + per-chain kernel)."""

    def __init__(self, n_channels, max_samples, psr_threshold=4.0, track_after=16, track_every=8,
                 device=0):
        h = C.c_void_p()
        self._h = None
        p = ChainParams(psr_threshold, track_after, track_every)
        _check(lib().lt_batch_create(C.byref(h), device, n_channels, int(max_samples), C.byref(p)),
               "lt_batch_create")
        self._h = h
        self.n_channels = n_channels
        self.max_samples = int(max_samples)
        self.rec_cap = record_capacity(max_samples)
        self._hf_cap = 0

    def close(self):
        if self._h and lib is not None:  # (module globals are None while the interpreter shuts down)
            lib().lt_batch_free(self._h)
            self._h = None

    __del__ = close

    def reset(self):
        _check(lib().lt_batch_reset(self._h), "lt_batch_reset")

    def set_psr_threshold(self, t):
        _check(lib().lt_batch_set_psr_threshold(self._h, t), "lt_batch_set_psr_threshold")

    def set_mode(self, mode):
        """1 = grid scan of the idle chains + per-chain kernel (default), 0 = per-chain kernel only."""
        _check(lib().lt_batch_set_mode(self._h, int(mode)), "lt_batch_set_mode")

    def mode(self):
        return int(lib().lt_batch_get_mode(self._h))

    def set_scan_groups(self, strips_per_workgroup):
        """Shape of the scan's workgroups (lt_batch_set_scan_groups): 0 = automatic, 1, 2 or 3 strips each."""
        _check(lib().lt_batch_set_scan_groups(self._h, int(strips_per_workgroup)), "lt_batch_set_scan_groups")

    def set_overlap(self, on):
        """Per-record kernels of a device run beside the next run's scan rounds (lt_batch_set_overlap)."""
        _check(lib().lt_batch_set_overlap(self._h, 1 if on else 0), "lt_batch_set_overlap")

    def set_window_cache(self, level):
        """Lag metrics formed in bulk: 0 = never, 1 = for the per-chain kernel, 2 = also for the scan rounds of small
        batches, 3 = 2 with the per-chain kernel's second pass only while chains ask for it, 4 (default) = 2 with the
        lag-metric streams of the chains that are not tracking (searches on any timing served from the cache: marginal
        SNR; the oracle's records, deterministic, but not the bits of levels 0..3) (lt_batch_set_window_cache)."""
        _check(lib().lt_batch_set_window_cache(self._h, int(level)), "lt_batch_set_window_cache")

    def cache_stats(self):
        """lt_batch_cache_stats of the last run, 7 values: (windows asked for, searches of pass 1 served from the cache,
        searches of pass 1 through the transforms, chains that waited for the cache, hits that moved such a chain off the
        grid its windows were listed on, windows an average was run through again from the cache, the same through the
        transforms)."""
        v = np.zeros(7, np.int64)
        _check(lib().lt_batch_cache_stats(self._h, v.ctypes.data_as(C.c_void_p)), "lt_batch_cache_stats")
        return tuple(int(x) for x in v)

    def set_phase_timing(self, level):
        """HIP events of a run: 2 = around it and between its stages, 1 = around it, 0 = none (lt_batch_set_phase_timing)."""
        _check(lib().lt_batch_set_phase_timing(self._h, int(level)), "lt_batch_set_phase_timing")

    def debug_hold_records(self, microseconds):
        """Diagnostics: every run's per-record kernels start this much later (lt_batch_debug_hold_records)."""
        _check(lib().lt_batch_debug_hold_records(self._h, int(microseconds)), "lt_batch_debug_hold_records")

    def set_scan_segments(self, first, later):
        _check(lib().lt_batch_set_scan_segments(self._h, int(first), int(later)), "lt_batch_set_scan_segments")

    def push_host(self, iq, end_of_stream=False):
        """Streaming: iq = the next [n_channels, n] complex64 chunk of the stream."""
        iq = np.ascontiguousarray(iq, dtype=np.complex64)
        if iq.ndim != 2 or iq.shape[0] != self.n_channels:
            raise ValueError("iq must be [n_channels, n_samples]")
        _check(lib().lt_batch_push(self._h, iq.ctypes.data_as(C.c_void_p), 0, iq.shape[1], int(bool(end_of_stream)), None),
               "lt_batch_push")

    def push_device(self, dev_ptr, n_samples, end_of_stream=False, stream=None):
        _check(lib().lt_batch_push(self._h, C.c_void_p(int(dev_ptr)), 1, int(n_samples), int(bool(end_of_stream)),
                                   C.c_void_p(int(stream)) if stream else None), "lt_batch_push")

    def enable_halfframes(self, max_per_chain):
        _check(lib().lt_batch_enable_halfframes(self._h, int(max_per_chain)), "lt_batch_enable_halfframes")
        self._hf_cap = int(max_per_chain)

    def run_host(self, iq):
        """iq: [n_channels, n] complex64 host array."""
        iq = np.ascontiguousarray(iq, dtype=np.complex64)
        if iq.ndim != 2 or iq.shape[0] != self.n_channels:
            raise ValueError("iq must be [n_channels, n_samples]")
        _check(lib().lt_batch_run(self._h, iq.ctypes.data_as(C.c_void_p), 0, iq.shape[1], None),
               "lt_batch_run")

    def run_device(self, dev_ptr, n_samples, stream=None):
        """dev_ptr: device address of [n_channels][n_samples] cf32; stream: hipStream_t int or None."""
        _check(lib().lt_batch_run(self._h, C.c_void_p(int(dev_ptr)), 1, int(n_samples),
                                  C.c_void_p(int(stream)) if stream else None), "lt_batch_run")

    def collect(self):
        """-> list over chains (channel*3 + n_id_2) of record arrays."""
        nch = self.n_channels * 3
        recs = np.zeros((nch, self.rec_cap), RECORD_DTYPE)
        counts = np.zeros(nch, np.int64)
        _check(lib().lt_batch_collect(self._h, recs.ctypes.data_as(C.c_void_p), self.rec_cap,
                                      counts.ctypes.data_as(C.c_void_p)), "lt_batch_collect")
        return [recs[c, : counts[c]] for c in range(nch)]

    def collect_previous(self):
        """Overlap mode: the records of the run before the last one (lt_batch_collect_previous)."""
        nch = self.n_channels * 3
        recs = np.zeros((nch, self.rec_cap), RECORD_DTYPE)
        counts = np.zeros(nch, np.int64)
        _check(lib().lt_batch_collect_previous(self._h, recs.ctypes.data_as(C.c_void_p), self.rec_cap,
                                               counts.ctypes.data_as(C.c_void_p)), "lt_batch_collect_previous")
        return [recs[c, : counts[c]] for c in range(nch)]

    def halfframes(self, chain):
        out = np.zeros((self._hf_cap, HALF_FRAME), np.complex64)
        n = C.c_int64()
        _check(lib().lt_batch_get_halfframes(self._h, chain, out.ctypes.data_as(C.c_void_p),
                                             self._hf_cap, C.byref(n)), "lt_batch_get_halfframes")
        return out[: n.value]

    def phase_ms(self, runs_back=0):
        """(plan + grid scan + resolve, 0, per-chain kernel, per-record kernels) device times in ms of the
        run `runs_back` runs ago (the last 8 runs are kept)."""
        ms = (C.c_float * 4)()
        _check(lib().lt_batch_phase_ms(self._h, int(runs_back), ms), "lt_batch_phase_ms")
        return tuple(float(v) for v in ms)

    def last_launch_count(self):
        """Kernel launches the last run enqueued (lt_batch_last_launch_count)."""
        n = C.c_int()
        _check(lib().lt_batch_last_launch_count(self._h, C.byref(n)), "lt_batch_last_launch_count")
        return n.value

    def last_kernel_ms(self):
        ms = C.c_float()
        _check(lib().lt_batch_last_kernel_ms(self._h, C.byref(ms)), "lt_batch_last_kernel_ms")
        return ms.value


class Decim:
    """rational_resampler_ccc(1, N) with GNU Radio's default taps, on the GPU."""

    def __init__(self, decimation, device=0):
        h = C.c_void_p()
        self._h = None
        _check(lib().lt_decim_create(C.byref(h), device, decimation), "lt_decim_create")
        self._h = h
        self.decimation = decimation

    def close(self):
        if self._h and lib is not None:  # (lib is None while the interpreter shuts down)
            lib().lt_decim_free(self._h)
            self._h = None

    __del__ = close

    def taps(self):
        n = lib().lt_decim_ntaps(self._h)
        t = np.empty(n, np.float32)
        _check(lib().lt_decim_get_taps(self._h, t.ctypes.data_as(C.c_void_p)), "lt_decim_get_taps")
        return t

    def run_host(self, x):
        x = np.ascontiguousarray(x, dtype=np.complex64)
        one = x.ndim == 1
        if one:
            x = x[None, :]
        y = np.empty((x.shape[0], x.shape[1] // self.decimation), np.complex64)
        _check(lib().lt_decim_run(self._h, x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p),
                                  x.shape[0], x.shape[1], 0, None), "lt_decim_run")
        return y[0] if one else y

    def run_device(self, x_ptr, y_ptr, n_channels, n_in, stream=None):
        _check(lib().lt_decim_run(self._h, C.c_void_p(int(x_ptr)), C.c_void_p(int(y_ptr)), n_channels,
                                  int(n_in), 1, C.c_void_p(int(stream)) if stream else None),
               "lt_decim_run")
