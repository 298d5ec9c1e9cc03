"""The 16-waves-per-CU form of the plain kernel (asdr_update_kernel_c16: 320-float LDS rows, <= 128 VGPRs, two-pass Hilbert FIR, even-sample
phase row, AGC table through L1 -- DESIGN.md 3.1 / 7) is OFF by default (measured slower than the four-wave workgroups: profiles/r05_c16_ab.txt)
and selected by the environment when the library makes its first launch -- so it is exercised in a process of its own: the parity, fuzz and
lane tests re-run with ASDR_C16=1 ASDR_C16_MIN_WAVES=1 (every direct one-block launch of an SSB-class group takes the kernel)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


@pytest.mark.gpu
def test_parity_suite_through_the_16_waves_per_cu_kernel(gpu):
    env = dict(os.environ, ASDR_C16="1", ASDR_C16_MIN_WAVES="1")
    out = subprocess.run([sys.executable, "-m", "pytest", "-q", "-m", "gpu", "-x", "-p", "no:cacheprovider",
                          os.path.join(ROOT, "tests", "test_gpu_parity.py"), os.path.join(ROOT, "tests", "test_gpu_fuzz.py"),
                          os.path.join(ROOT, "tests", "test_gpu_lanes.py")],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-1000:]
    assert " passed" in out.stdout
