"""Build-time properties of the HIP kernels that the measured performance depends on (profiles/README.md):
no VGPR spills, the register and LDS budgets that give 12 waves/CU for the main instantiation.  hipcc cross-compiles
without a GPU; the check reads the compiler's own resource remarks."""
import os
import re
import subprocess

import pytest

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
CSRC = os.path.join(ROOT, "audiosdr_amd", "csrc")


def _resources(src):
    from audiosdr_amd import build as b
    flags = [f for f in b.FLAGS if f not in ("-shared", "-fPIC")]
    cmd = [b.hipcc(), "--offload-arch=" + b.ARCH] + flags + ["--cuda-device-only", "-c", src, "-o", os.devnull,
                                                             "-Rpass-analysis=kernel-resource-usage"]
    out = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    res, cur = {}, None
    for line in out.stderr.splitlines():
        m = re.search(r"remark: (?:Function Name: (\S+)|\s*(VGPRs|VGPRs Spill|SGPRs Spill|LDS Size \[bytes/block\]|ScratchSize \[bytes/lane\]): (\d+))", line)
        if not m:
            continue
        if m.group(1):
            cur = res.setdefault(m.group(1), {})
        elif cur is not None:
            cur[m.group(2)] = int(m.group(3))
    return res


@pytest.mark.parametrize("src", ["asdr_kernels.hip", "asdr_front.hip"])
def test_kernels_do_not_spill(src):
    res = _resources(src)
    assert res, "no resource remarks"
    for name, r in res.items():
        assert r.get("VGPRs Spill", 0) == 0, (name, r)


def test_main_kernel_keeps_its_occupancy_budget():
    """12 waves/CU for the plain instantiation needs <= 168 VGPRs (3 waves/SIMD) and <= 12,800 B of LDS per wave (the LDS
    allocation granule on gfx950 puts 12,944 B at 11 waves again: profiles/README.md (m))."""
    r = _resources("asdr_kernels.hip")["asdr_update_kernel"]
    assert r["VGPRs"] <= 168 and r["LDS Size [bytes/block]"] <= 12800, r


def test_the_16_waves_per_cu_kernel_fits_its_budget():
    """asdr_update_kernel_c16 exists to run FOUR waves per SIMD: <= 128 VGPRs without spills and <= 10,240 B of LDS per wave (16 workgroups x
    10,240 B = the CU's 160 KB)."""
    r = _resources("asdr_kernels.hip")["asdr_update_kernel_c16"]
    assert r["VGPRs"] <= 128 and r.get("VGPRs Spill", 0) == 0 and r["LDS Size [bytes/block]"] <= 10240, r
