import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# fixture name -> (file, decimation to 1.92 Msps, expected cell dict fields)
# python/qa_downlink_trigger_c.py:74,107,142,177 (rates) and :93-98,128-133,163-168,198-203 (dicts)
FIXTURES = {
    "6prb": ("lte_frame_6prb_cellid_123", 1, dict(cell_id=123, nof_prb=6)),
    "25prb": ("lte_frame_25prb_cellid_124", 4, dict(cell_id=124, nof_prb=25)),
    "50prb": ("lte_frame_50prb_cellid_125", 8, dict(cell_id=125, nof_prb=50)),
    "100prb": ("lte_frame_100prb_cellid_369", 16, dict(cell_id=369, nof_prb=100)),
}


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_frame(name):
    fname, decim, expect = FIXTURES[name]
    x = np.fromfile(os.path.join(GOLDEN, "test_frames", fname), dtype=np.complex64)
    return x, decim, expect


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def frames_192(oracle):
    """The four fixtures, looped x3 at their native rate and decimated to 1.92 Msps with the oracle
    decimator -> dict name -> complex64[57600] (three radio frames, filter transient included)."""
    out = {}
    for name in FIXTURES:
        x, decim, _ = load_frame(name)
        out[name] = oracle.decimate(np.tile(x, 3), decim)
    return out


@pytest.fixture(scope="session")
def lt():
    try:
        import torch  # noqa: F401  (before the library: see _binding.lib() on the two HIP runtimes)
    except ImportError:
        pass
    import ltetrigger_amd
    ltetrigger_amd.lib()
    return ltetrigger_amd
