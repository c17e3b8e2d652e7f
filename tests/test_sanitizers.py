"""CPU-side sanitizer runs (the reference has only an assert that Release builds compile out, lib/pss_impl.cc:191).

`make -C gr-ltetrigger_amd asan tsan` builds
  * the host library (blocks, mini runtime, PBCH decoder, ctypes C interface) and the CPU oracle with AddressSanitizer +
    UndefinedBehaviorSanitizer -- the CPU suite's host-side tests are then run once more against THOSE builds (a child
    pytest with the sanitizer runtimes preloaded, the libraries selected through LT_HOST_LIB / LT_ORACLE_LIB);
  * tests/hostcheck/runtime_tsan.cc with ThreadSanitizer: two threads posting track / drop into a cellstore while a third
    polls it (lib/cellstore_impl.cc:60-105), and a flowgraph under the thread-per-block scheduler against the
    single-threaded one.
No GPU sanitizers exist on this pool; the device side is held by the parity tests.
"""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "gr-ltetrigger_amd")
SUITES = ["tests/test_mib.py", "tests/test_oracle_fixtures.py", "tests/test_synth_cells.py", "tests/test_strip_model.py",
          "tests/test_host_blocks.py"]


def _runtime(name):
    p = subprocess.run(["gcc", f"-print-file-name={name}"], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


@pytest.fixture(scope="module")
def built():
    if not shutil.which("g++") or not os.path.exists(os.path.join(PKG, "libltetrigger_hip.so")):
        pytest.skip("needs g++ and the built C-ABI library (the host library links against it)")
    subprocess.check_call(["make", "-C", PKG, "asan", "tsan"], stdout=subprocess.DEVNULL)
    return {"host": os.path.join(PKG, "build", "asan", "libgnuradio-ltetrigger-amd.so"),
            "oracle": os.path.join(PKG, "build", "asan", "liblte_oracle.so"),
            "tsan": os.path.join(PKG, "build", "tsan", "runtime_tsan")}


def test_host_side_suites_under_asan_and_ubsan(built):
    asan, ubsan = _runtime("libasan.so"), _runtime("libubsan.so")
    if not asan or not ubsan:
        pytest.skip("gcc's sanitizer runtimes are not installed")
    env = dict(os.environ, LT_HOST_LIB=built["host"], LT_ORACLE_LIB=built["oracle"], LD_PRELOAD=f"{asan}:{ubsan}",
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    # the sanitized builds are the ones a process with this environment maps
    probe = ("import ltetrigger_amd.host as h; h.hlib(); from oracle import oracle as O; O.lib(); "
             "print([l.split()[-1] for l in open('/proc/self/maps') if 'build/asan' in l])")
    p = subprocess.run([sys.executable, "-c", probe], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    assert "build/asan/libgnuradio-ltetrigger-amd.so" in p.stdout and "build/asan/liblte_oracle.so" in p.stdout, p.stdout
    p = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider"] + SUITES,
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    out = p.stdout + p.stderr
    assert p.returncode == 0, out[-4000:]
    assert "AddressSanitizer" not in out and "runtime error:" not in out, out[-4000:]
    assert " passed" in p.stdout


def test_cellstore_and_threaded_scheduler_under_tsan(built):
    p = subprocess.run([built["tsan"]], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, TSAN_OPTIONS="halt_on_error=0:second_deadlock_stack=1"))
    assert p.returncode == 0, (p.stdout + p.stderr)[-4000:]
    assert "ThreadSanitizer" not in p.stderr, p.stderr[-4000:]
    assert "runtime_tsan ok" in p.stdout
