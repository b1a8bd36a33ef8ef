"""AddressSanitizer + UndefinedBehaviorSanitizer runs of the host code (GPU sanitizers are not available on this pool):
the CPU oracle over four small scenes with exactly-sized buffers (oracle/san_driver.c) and the library's host-side code
that needs no GPU -- view layouts, workspace sizes, the merge walk (csrc/san_host.cpp; every -fsanitize= sits behind
-Xarch_host).  Both are built here by their Makefiles and must run clean."""
import os
import shutil
import subprocess
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _make(directory, target, timeout):
    r = subprocess.run(["make", "-C", os.path.join(ROOT, directory), target, "-j4"], capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
    return r.stdout


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_oracle_runs_clean_under_asan_ubsan():
    out = _make("oracle", "san", 600)
    assert out.count("checksum") == 4


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_library_host_code_runs_clean_under_asan_ubsan():
    out = _make(os.path.join("gaussian-splatting_amd", "csrc"), "san_host", 1500)
    assert "san_host: ok" in out
