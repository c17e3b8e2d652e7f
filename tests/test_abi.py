"""The C-ABI library loads on a GPU-less host and exports every symbol include/ltetrigger_hip.h
declares; compute entry points fail loudly (no CPU fallback)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def test_library_exports_every_declared_symbol(lt):
    hdr = open(os.path.join(ROOT, "include", "ltetrigger_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(lt_[a-zA-Z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(lt.EXPORTS), declared ^ set(lt.EXPORTS)
    lib = ctypes.CDLL(lt.LIB_PATH)
    for sym in sorted(declared):
        assert hasattr(lib, sym), f"{sym} not exported"


def test_record_layout_matches_header(lt, oracle):
    assert lt.RECORD_DTYPE.itemsize == 96
    assert lt.RECORD_DTYPE == oracle.RECORD_DTYPE
    assert oracle.lib().orc_sizeof_record() == 96


def test_compute_calls_fail_loudly_without_gpu(lt):
    if lt.lib().lt_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(lt.LtError, match="no CPU fallback"):
        lt.Pss(0)
    with pytest.raises(lt.LtError, match="no CPU fallback"):
        lt.Batch(1, 9600)
    with pytest.raises(lt.LtError):
        lt.Decim(8)


def test_argument_validation(lt):
    L = lt.lib()
    h = ctypes.c_void_p()
    assert L.lt_pss_init(ctypes.byref(h), 0, 1234, 128) == -2      # SRSLTE_ERROR_INVALID_INPUTS
    assert L.lt_pss_init(None, 0, 9600, 128) == -2
    # a call consumes >= 8640 samples; a chunk may also finish calls left from the previous one (<= 19456 samples back)
    assert L.lt_batch_record_capacity(1_920_000) == (1_920_000 + 19456) // 8640 + 2
    assert b"gfx950" in L.lt_version()
