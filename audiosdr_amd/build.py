"""Compile libasdr_hip.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU.  Flags that matter for parity:
  -ffp-contract=off     no FMA contraction: every float op is separately rounded, as in the
                        reference's arithmetic (SURVEY.md 7 "FMA contraction")
  -fno-slp-vectorize    keeps the sliding-window FIR out of v_pk_* + realignment moves
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libasdr_hip.so")
SOURCES = ["asdr_kernels.hip", "asdr_host.cpp", "asdr_front.hip", "asdr_front_host.cpp"]
DEPS = SOURCES + ["asdr_device.h", "asdr_fir.h", "asdr_tables.h", "asdr_front_device.h", "asdr_front_tables.h",
                  os.path.join("..", "..", "include", "asdr.h"), os.path.join("..", "..", "include", "asdr_front.h")]
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fno-fast-math", "-fPIC", "-shared",
         "-Wall", "-Wno-unused-function", "-Wno-unused-value"]


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def source_sha256():
    """sha256 over the library's sources, headers and build flags (path-independent, unlike the binary's hash: hipcc's output
    depends on where the tree lives): stamps measurement files with the code they were taken from."""
    import hashlib
    h = hashlib.sha256()
    h.update((" ".join(FLAGS) + " " + ARCH).encode())
    for d in DEPS:
        with open(os.path.join(CSRC, d), "rb") as f:
            h.update(d.encode()); h.update(f.read())
    return h.hexdigest()


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


def build(force=False, extra_flags=(), out=LIB, verbose=False):
    if not force and out == LIB and not is_stale():
        return out
    cmd = [hipcc(), "--offload-arch=" + ARCH] + FLAGS + list(extra_flags) + SOURCES + ["-o", out]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd, cwd=CSRC)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
