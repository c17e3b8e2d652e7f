"""Python face of the host blocks (ctypes over host/capi.h) -- the role SWIG plays in the
reference (swig/ltetrigger_swig.i:19-29): make blocks, wire a flowgraph, run it, read results."""
import ctypes as C
import json
import os

import numpy as np

from . import _binding

_HERE = os.path.dirname(os.path.abspath(__file__))
# (LT_HOST_LIB: another build of the same library -- the AddressSanitizer one of `make asan`, tests/test_sanitizers.py)
HOST_LIB_PATH = os.environ.get("LT_HOST_LIB") or os.path.join(_HERE, "libgnuradio-ltetrigger-amd.so")

_hlib = None


class Mib(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("nof_prb", "nof_ports", "phich_length_ext", "phich_resources", "sfn", "frame_idx")]


def hlib():
    global _hlib
    if _hlib is not None:
        return _hlib
    _binding.lib()  # the C-ABI library must be loadable first (rpath $ORIGIN finds it as well)
    if not os.path.exists(HOST_LIB_PATH):
        raise _binding.LtError(f"{HOST_LIB_PATH} is missing: build it with `make -C {_HERE}`")
    L = C.CDLL(HOST_LIB_PATH)
    vp, i32, i64, f32, cp = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_char_p
    L.ltr_tb_create.restype = vp
    L.ltr_tb_free.argtypes = [vp]
    L.ltr_last_error.restype = cp
    L.ltr_add_vector_source.argtypes = [vp, vp, i64, i32]
    L.ltr_add_file_source.argtypes = [vp, cp, i32]
    L.ltr_add_head.argtypes = [vp, i64]
    L.ltr_add_resampler.argtypes = [vp, i32]
    L.ltr_add_pss.argtypes = [vp, i32, f32, i32, i32]
    L.ltr_add_sss.argtypes = [vp, i32]
    L.ltr_add_mib.argtypes = [vp, i32]
    for f in ("ltr_add_cellstore", "ltr_add_message_debug", "ltr_add_null_sink", "ltr_add_probe", "ltr_run"):
        getattr(L, f).argtypes = [vp]
    L.ltr_run_timeout.argtypes = [vp, C.c_double]
    L.ltr_run_threaded.argtypes = [vp, C.c_double]
    L.ltr_connect.argtypes = [vp, i32, i32]
    L.ltr_msg_connect.argtypes = [vp, i32, cp, i32, cp]
    for f in ("ltr_pss_max_psr", "ltr_pss_mean_psr", "ltr_pss_mean_cfo", "ltr_pss_psr_threshold",
              "ltr_pss_tracking_score"):
        getattr(L, f).argtypes = [vp, i32]
        getattr(L, f).restype = f32
    L.ltr_pss_set_psr_threshold.argtypes = [vp, i32, f32]
    L.ltr_msgdebug_count.argtypes = [vp, i32]
    L.ltr_msgdebug_json.argtypes = [vp, i32, i32, cp, i32]
    L.ltr_cellstore_tracking.argtypes = [vp, i32]
    L.ltr_cellstore_cells_json.argtypes = [vp, i32, cp, i32]
    L.ltr_cellstore_latest_json.argtypes = [vp, i32, cp, i32]
    L.ltr_probe_nitems.argtypes = [vp, i32]
    L.ltr_probe_nitems.restype = i64
    L.ltr_probe_items.argtypes = [vp, i32, vp, i64]
    L.ltr_probe_items.restype = i64
    L.ltr_probe_tags_json.argtypes = [vp, i32, cp, i32]
    L.ltr_decode_mib.argtypes = [vp, i32, i32, i32, C.POINTER(Mib)]
    _hlib = L
    return L


def _err():
    return hlib().ltr_last_error().decode(errors="replace")


def _id(rc, what):
    if rc < 0:
        raise RuntimeError(f"{what}: {_err()}")
    return rc


def _json_call(fn, *args):
    size = 1 << 16
    while True:
        buf = C.create_string_buffer(size)
        rc = fn(*args, buf, size)
        if rc >= 0:
            return json.loads(buf.value.decode())
        if rc == -1:
            raise RuntimeError(_err())
        size = -rc + 16


PHICH_R = ["1/6", "1/2", "1", "2"]


def decode_mib(subframe1920, cell_id, cp_ext=False, nof_ports_hint=0):
    """PBCH/MIB decode of one subframe (host/pbch.cc); None when no CRC matches."""
    x = np.ascontiguousarray(subframe1920, dtype=np.complex64)
    if x.size < 1920:
        raise ValueError("need 1920 samples")
    m = Mib()
    rc = hlib().ltr_decode_mib(x.ctypes.data_as(C.c_void_p), int(cell_id), int(bool(cp_ext)),
                               int(nof_ports_hint), C.byref(m))
    if rc < 0:
        raise RuntimeError(_err())
    if rc == 0:
        return None
    return dict(nof_prb=m.nof_prb, nof_tx_ports=m.nof_ports,
                phich_len="Extended" if m.phich_length_ext else "Normal",
                nof_phich_resources=PHICH_R[m.phich_resources & 3], sfn_offset=m.sfn,
                frame_idx=m.frame_idx)


class Block:
    def __init__(self, tb, bid):
        self.tb, self.id = tb, bid


class PssBlock(Block):
    """ltetrigger.pss getters (include/ltetrigger/pss.h:72-87)."""

    def max_psr(self):
        return hlib().ltr_pss_max_psr(self.tb._h, self.id)

    def mean_psr(self):
        return hlib().ltr_pss_mean_psr(self.tb._h, self.id)

    def mean_cfo(self):
        return hlib().ltr_pss_mean_cfo(self.tb._h, self.id)

    def psr_threshold(self):
        return hlib().ltr_pss_psr_threshold(self.tb._h, self.id)

    def tracking_score(self):
        return hlib().ltr_pss_tracking_score(self.tb._h, self.id)

    def set_psr_threshold(self, t):
        _id(hlib().ltr_pss_set_psr_threshold(self.tb._h, self.id, t), "set_psr_threshold")


class MessageDebug(Block):
    def num_messages(self):
        return _id(hlib().ltr_msgdebug_count(self.tb._h, self.id), "num_messages")

    def get_message(self, i):
        return _json_call(hlib().ltr_msgdebug_json, self.tb._h, self.id, i)


class CellStore(Block):
    """ltetrigger.cellstore (include/ltetrigger/cellstore.h:56-65)."""

    def tracking(self):
        return bool(_id(hlib().ltr_cellstore_tracking(self.tb._h, self.id), "tracking"))

    def cells(self):
        return _json_call(hlib().ltr_cellstore_cells_json, self.tb._h, self.id)

    def latest_cell(self):
        return _json_call(hlib().ltr_cellstore_latest_json, self.tb._h, self.id)


class Probe(Block):
    def items(self):
        n = hlib().ltr_probe_nitems(self.tb._h, self.id)
        out = np.empty(max(n, 0), np.complex64)
        if n > 0:
            hlib().ltr_probe_items(self.tb._h, self.id, out.ctypes.data_as(C.c_void_p), n)
        return out

    def tags(self):
        return _json_call(hlib().ltr_probe_tags_json, self.tb._h, self.id)


class TopBlock:
    """gr.top_block stand-in on the in-repo runtime."""

    def __init__(self):
        self._h = hlib().ltr_tb_create()

    def __del__(self):
        if getattr(self, "_h", None):
            hlib().ltr_tb_free(self._h)
            self._h = None

    # block factories -------------------------------------------------
    def vector_source(self, data, repeat=False):
        x = np.ascontiguousarray(data, dtype=np.complex64)
        return Block(self, _id(hlib().ltr_add_vector_source(self._h, x.ctypes.data_as(C.c_void_p), x.size,
                                                            int(repeat)), "vector_source"))

    def file_source(self, path, repeat=False):
        return Block(self, _id(hlib().ltr_add_file_source(self._h, os.fsencode(path), int(repeat)), "file_source"))

    def head(self, nitems):
        return Block(self, _id(hlib().ltr_add_head(self._h, int(nitems)), "head"))

    def rational_resampler_ccc(self, interpolation, decimation):
        if interpolation != 1:
            raise ValueError("only interpolation=1")
        return Block(self, _id(hlib().ltr_add_resampler(self._h, int(decimation)), "rational_resampler_ccc"))

    def pss(self, N_id_2, psr_threshold, track_after=16, track_every=8):
        return PssBlock(self, _id(hlib().ltr_add_pss(self._h, N_id_2, psr_threshold, track_after, track_every),
                                  "Error initializing PSS"))

    def sss(self, N_id_2):
        return Block(self, _id(hlib().ltr_add_sss(self._h, N_id_2), "Error initializing SSS"))

    def mib(self, exit_on_success=False):
        return Block(self, _id(hlib().ltr_add_mib(self._h, int(exit_on_success)), "mib"))

    def cellstore(self):
        return CellStore(self, _id(hlib().ltr_add_cellstore(self._h), "cellstore"))

    def message_debug(self):
        return MessageDebug(self, _id(hlib().ltr_add_message_debug(self._h), "message_debug"))

    def null_sink(self):
        return Block(self, _id(hlib().ltr_add_null_sink(self._h), "null_sink"))

    def probe(self):
        return Probe(self, _id(hlib().ltr_add_probe(self._h), "probe"))

    # wiring -----------------------------------------------------------
    def connect(self, *blocks):
        for a, b in zip(blocks[:-1], blocks[1:]):
            if hasattr(b, "_hier_connect_input"):
                b._hier_connect_input(self, a)
            else:
                _id(hlib().ltr_connect(self._h, a.id, b.id), "connect")

    def msg_connect(self, src, src_port, dst, dst_port):
        if hasattr(src, "_hier_msg_sources"):
            for s in src._hier_msg_sources(src_port):
                _id(hlib().ltr_msg_connect(self._h, s.id, src_port.encode(), dst.id, dst_port.encode()), "msg_connect")
        else:
            _id(hlib().ltr_msg_connect(self._h, src.id, src_port.encode(), dst.id, dst_port.encode()), "msg_connect")

    def run(self, time_out=-1.0, threaded=False):
        """Run to completion; time_out >= 0 bounds the wall time (tb.start(); sleep; tb.stop()).  threaded: one thread per
        block, as GNU Radio's scheduler runs the reference (ctypes releases the GIL: other Python threads may poll the
        getters meanwhile)."""
        if threaded:
            _id(hlib().ltr_run_threaded(self._h, float(time_out)), "run")
        else:
            _id(hlib().ltr_run_timeout(self._h, float(time_out)), "run")
