"""CPU check of the FFT-1024 (csrc/lt_fft.h): the lane-level code is compiled for the host
and emulated lane by lane (tests/hostcheck/fft_host_check.cpp), then compared with numpy."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from conftest import ROOT

CLANG = "/opt/rocm/lib/llvm/bin/clang++"


@pytest.mark.skipif(not os.path.exists(CLANG), reason="needs the ROCm clang (vector extensions)")
def test_lane_level_fft_matches_numpy(tmp_path):
    exe = str(tmp_path / "fft_host_check")
    src = os.path.join(ROOT, "tests", "hostcheck", "fft_host_check.cpp")
    subprocess.check_call([CLANG, "-O2", "-std=c++17", "-ffp-contract=off", "-o", exe, src])
    rng = np.random.default_rng(0)
    for trial in range(3):
        x = (rng.standard_normal(1024) + 1j * rng.standard_normal(1024)).astype(np.complex64)
        if trial == 2:
            x[:] = 0
            x[5] = 1
        out = subprocess.run([exe], input=x.tobytes(), capture_output=True, check=True).stdout
        r = np.frombuffer(out, dtype=np.complex64)
        fwd, inv = r[:1024], r[1024:]
        ref = np.fft.fft(x.astype(np.complex128))
        assert np.max(np.abs(fwd - ref)) <= 5e-7 * max(np.max(np.abs(ref)), 1e-30)
        assert np.max(np.abs(inv / 1024 - x)) <= 3e-6
