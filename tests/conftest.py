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


# Filled by pytest_sessionstart when the GPU tests are selected: the finished `python bench.py --gpus 2` child
# (returncode, stdout, stderr, steps) for tests/test_gpu_bench_configs.py::test_two_ranks_share_one_gpu.
REHEARSAL = {}


def pytest_sessionstart(session):
    """`-m gpu`: the rehearsal of the N > 1 path -- two ranks of bench.py sharing this box's one GPU -- runs HERE, to its
    end, before any test of this process has touched the GPU: a process that has initialised the GPU must not start
    programs that use it (on the GPU boxes that takes the machine down).  torch.cuda.device_count() does not initialise
    it.  LT_SKIP_REHEARSAL=1 skips (the test then skips too)."""
    expr = (session.config.getoption("markexpr", "") or "").replace(" ", "")
    if "gpu" not in expr or "notgpu" in expr or os.environ.get("LT_SKIP_REHEARSAL"):
        return
    if getattr(session.config, "workerinput", None) is not None:  # (an xdist worker: the controller has done it)
        return
    try:
        import torch
        if torch.cuda.device_count() < 1:
            return
    except Exception:
        return
    import subprocess
    steps = 3
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", str(steps), "--warmup", "2",
           "--no-end-to-end", "--no-overlap-leg", "--no-steady-leg"]
    for attempt in range(2):  # (once more if the launcher's rendezvous failed: bench.py picks a fresh port per start)
        try:
            p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
            REHEARSAL.update(returncode=p.returncode, stdout=p.stdout, stderr=p.stderr, steps=steps, attempts=attempt + 1)
        except subprocess.TimeoutExpired as e:
            REHEARSAL.update(returncode=-1, stdout=str(e.stdout or ""), stderr="timeout: " + str(e.stderr or ""), steps=steps,
                             attempts=attempt + 1)
            break
        if p.returncode == 0:
            break


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
