#!/usr/bin/env python3
"""Per-kernel register / LDS / spill table of the HIP sources (the compiler's own resource remarks; cross-compiles here).
  python tools/kernel_resources.py [extra hipcc flags ...]"""
import os
import re
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from audiosdr_amd import build as b  # noqa: E402

KEYS = ["VGPRs", "AGPRs", "SGPRs", "VGPRs Spill", "SGPRs Spill", "ScratchSize [bytes/lane]", "LDS Size [bytes/block]", "Occupancy [waves/SIMD]"]


def resources(src, extra=()):
    flags = [f for f in b.FLAGS if f not in ("-shared", "-fPIC")]
    cmd = [b.hipcc(), "--offload-arch=" + b.ARCH] + flags + list(extra) + ["--cuda-device-only", "-c", src, "-o", os.devnull,
                                                                            "-Rpass-analysis=kernel-resource-usage"]
    out = subprocess.run(cmd, cwd=b.CSRC, capture_output=True, text=True)
    if out.returncode != 0:
        sys.exit(out.stderr[-3000:])
    res, cur = {}, None
    for line in out.stderr.splitlines():
        m = re.search(r"remark: (?:Function Name: (\S+)|\s*([A-Za-z \[\]/]+): (\d+))", line)
        if not m:
            continue
        if m.group(1):
            cur = res.setdefault(m.group(1), {})
        elif cur is not None:
            cur[m.group(2).strip()] = int(m.group(3))
    return res


if __name__ == "__main__":
    extra = sys.argv[1:]
    for src in ("asdr_kernels.hip", "asdr_front.hip"):
        for name, r in resources(src, extra).items():
            print("%-40s " % name + "  ".join("%s=%s" % (k.split(" [")[0].replace(" ", ""), r.get(k, "-")) for k in KEYS))
