"""Runs the C++ host-mirror test binary (the reference's inline tests restated in C++ over the C ABI)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_cpp_host_mirror():
    exe = os.path.join(ROOT, "tests", "cpp", "test_host_mirror")
    # always through make: a binary older than zolt_host.hpp or the header must not be the one that is tested
    subprocess.check_call(["make", "-C", os.path.dirname(exe), "test_host_mirror"])
    res = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    print(res.stdout)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "0 failures" in res.stdout
