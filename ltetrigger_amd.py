"""Import shim: ``import ltetrigger_amd`` -> the package in ``gr-ltetrigger_amd/``."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
sys.modules[__name__] = importlib.import_module("gr-ltetrigger_amd")
