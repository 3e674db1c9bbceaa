"""not gpu: the host build of ansel_amd/csrc/devmath.h (the restated glibc powf/log2f/exp2f/expf/atan2f/hypotf)
returns the bits of this machine's libm on every argument tried, including arbitrary bit patterns."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("devmath") / "devmath_host")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-mfma", "-ffp-contract=off",
                           "-I" + os.path.join(ROOT, "ansel_amd", "csrc"),
                           os.path.join(ROOT, "tests", "native", "devmath_host.cpp"), "-o", exe, "-lm"])
    return exe


@pytest.mark.parametrize("fn", ["powf", "log2f", "logf", "exp2f", "expf", "atan2f", "hypotf", "sinf", "cosf"])
def test_restated_libm_is_bit_exact(harness, fn):
    out = subprocess.run([harness, fn, "6000000", "7"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    bad, n = out.stdout.split()
    assert bad == "0" and int(n) == 6000000
