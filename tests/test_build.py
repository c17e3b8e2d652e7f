"""Compile-time properties of the gfx950 kernels (hipcc cross-compiles without a GPU): no product kernel spills
to scratch, and the scan kernel keeps the shape its design counts on (three waves per SIMD, one workgroup's LDS)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "gr-ltetrigger_amd")


@pytest.fixture(scope="module")
def resources():
    if not shutil.which("hipcc"):
        pytest.skip("hipcc not on PATH")
    subprocess.check_call(["make", "-C", PKG, "asm"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    text = open(os.path.join(PKG, "build", "lt_kernels.resource.txt")).read()
    out = {}
    for block in text.split("Function Name: ")[1:]:
        name = block.split()[0]
        get = lambda key: int(re.search(key + r": (\d+)", block).group(1))
        out[name] = {"vgprs": get(r"VGPRs"), "scratch": get(r"ScratchSize \[bytes/lane\]"), "lds": get(r"LDS Size \[bytes/block\]")}
    return out


def test_no_kernel_uses_scratch(resources):
    assert len(resources) >= 18
    bad = {k: v["scratch"] for k, v in resources.items() if v["scratch"] != 0}
    assert not bad, f"kernels with scratch: {bad}"


def test_scan_kernel_shape(resources):
    scan = {k: v for k, v in resources.items() if "lt_scan_kernel" in k}
    assert len(scan) == 3, "one instance per number of strips per workgroup (1, 2, 3)"
    for k, v in scan.items():
        assert v["vgprs"] <= 168, f"{k}: at most three waves of a workgroup share a SIMD: 168 VGPRs"
        assert v["lds"] <= 160 * 1024, f"{k}: a workgroup's LDS on gfx950"
    assert min(v["lds"] for v in scan.values()) <= 80 * 1024, "the one-strip shape fits a CU twice"
