"""The PLL kernel evaluates approx_atan2 (AudioSDR.h:384-408) branch-free; round 5 shortened that form (sign-bit half_pi, the x == 0 branch
folded into the general one).  tools/check_atan2_forms.c holds both forms in C: equal bit for bit -- NaNs included -- on every pair of 18
special values and on 2 x 10^7 random operand pairs (4 x 10^8 when run by hand)."""
import os
import subprocess

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def test_round5_atan2_form_equals_the_round4_form(tmp_path):
    exe = str(tmp_path / "check_atan2")
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-msse2", "-mfpmath=sse", os.path.join(ROOT, "tools", "check_atan2_forms.c"), "-lm", "-o", exe], check=True)
    out = subprocess.run([exe, "20000000"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:]
    assert "non-NaN mismatches 0, NaN-bit mismatches 0" in out.stdout, out.stdout[-500:]
