"""The GNU Radio LTETRIGGER module on MI355X: pss / sss / mib / cellstore blocks and the downlink_trigger_c hier block.

This package is what gets installed as ``ltetrigger`` beside a real GNU Radio (root CMakeLists.txt, Gnuradio_FOUND).  It
has NOT been imported anywhere in this repository's environment -- there is no GNU Radio here; the in-repo counterpart
that runs without it is the ``ltetrigger_amd`` package (gr-ltetrigger_amd/__init__.py).
"""
try:
    from . import ltetrigger_python as blocks_module  # pybind11 bindings, GNU Radio >= 3.9
    from .ltetrigger_python import *  # noqa: F401,F403
except ImportError:
    # GNU Radio 3.7: the reference's SWIG interface file on ../host/ltetrigger/*.h
    import ltetrigger_swig as blocks_module
    from ltetrigger_swig import *  # noqa: F401,F403

from .downlink_trigger_c import downlink_trigger_c, MIN_PSR_THRESHOLD  # noqa: E402,F401
