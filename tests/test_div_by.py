"""The exact quotient with a hoisted reciprocal (div_by(), ansel_amd/csrc/bilat.hip) against the host's `/`:
tools/div_by_check.c, 2e7 random cases x 5 reciprocal seeds (the full run of 4e8 is in the tool's header)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_hoisted_reciprocal_quotient_is_correctly_rounded(tmp_path):
    exe = str(tmp_path / "div_by_check")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-o", exe, os.path.join(ROOT, "tools", "div_by_check.c"), "-lm"])
    out = subprocess.run([exe, "20000000"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.strip().startswith("0 bad of 100000000"), out.stdout
