"""AddressSanitizer + UndefinedBehaviorSanitizer over the CPU oracle and the host build of the device integrator
(`make -C tests/c asan`, tests/c/sanitize_driver.cpp).  GPU sanitizers are not available on the test pool, so the
kernel logic is sanitised where it can be: compiled for the host from the very same header the kernels are built from."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None or shutil.which("make") is None, reason="needs g++ and make")
def test_oracle_and_host_kernel_logic_are_clean_under_asan_and_ubsan():
    out = subprocess.run(["make", "-s", "-j4", "-C", os.path.join(ROOT, "tests", "c"), "asan"], capture_output=True, text=True, timeout=900)
    tail = (out.stdout + out.stderr)[-3000:]
    assert out.returncode == 0, tail
    assert "ERROR: AddressSanitizer" not in tail and "runtime error" not in tail, tail
    assert out.stdout.strip().endswith("OK"), tail
