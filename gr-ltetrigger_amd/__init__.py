"""MI355X-native LTE downlink cell search (PSS/SSS) behind NTIA/gr-ltetrigger's block surface.

Importable as ``ltetrigger_amd`` through the shim at the repository root (the directory name
carries a hyphen).  The compute path is the gfx950 shared library built from ``csrc/``; there
is no CPU implementation in this package.
"""
from ._binding import (  # noqa: F401
    Batch, Cfo, ChainParams, Decim, LtError, Pss, Sync, RECORD_DTYPE, EXPORTS, LIB_PATH,
    F_SEARCHED, F_OVER, F_TRACKING, F_EMITTED, F_TAG_LOST, F_TAG_CELL, F_CP_EXT, F_SSS_RAN,
    HALF_FRAME, SLOT, SYMBOL_SZ, CONV_LEN, NLAGS, CP_NORM, CP_EXT,
    last_error, lib, record_capacity,
)
from . import host  # noqa: E402,F401
from .host import TopBlock, decode_mib  # noqa: E402,F401
from .downlink_trigger_c import downlink_trigger_c, MIN_PSR_THRESHOLD  # noqa: E402,F401
from .batch_trigger import BatchTrigger  # noqa: E402,F401
from . import sharding  # noqa: E402,F401
