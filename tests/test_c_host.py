"""A plain C99 host program over the C ABI (tests/host_c/receiver_bank.c): `gcc -std=c99 -pedantic` must accept include/asdr.h as C,
and the program -- a sharded batch of WSPR receivers configured through the reference's method names, host rows in page-locked
memory, one asdr_update() per audio period, the oracle linked in-process as the checker -- must report zero differing samples and
zero differing status words (BASELINE.json north_star: "C host code over a thin C-ABI")."""
import os
import subprocess

import pytest

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
SRC = os.path.join(ROOT, "tests", "host_c", "receiver_bank.c")
EXE = os.path.join(ROOT, "tests", "host_c", "receiver_bank")


def _build(A, ao):
    lib_dir = os.path.dirname(A.library_path())
    orc_dir = os.path.join(ROOT, "oracle")
    ao.build()
    cmd = ["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-O1", "-I", os.path.join(ROOT, "include"), "-I", orc_dir, SRC, "-o", EXE,
           "-L", lib_dir, "-lasdr_hip", "-Wl,-rpath," + lib_dir, "-L", orc_dir, "-lasdr_oracle", "-Wl,-rpath," + orc_dir,
           "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-lm"]
    subprocess.check_call(cmd)


def test_the_header_is_c_and_the_host_builds(A, ao):
    _build(A, ao)
    out = subprocess.run([EXE, "-1", "8", "1", "1"], capture_output=True, text=True)     # no device: creation must fail loudly, not fall back
    assert out.returncode == 2 and "needs a HIP device" in (out.stdout + out.stderr), out.stdout + out.stderr


@pytest.mark.gpu
def test_c_host_receiver_bank_is_bit_exact(gpu, ao):
    _build(gpu, ao)
    # the second: 1,500 receivers x 8-block periods (block pipeline per shard); the third: ordinary malloc'ed rows (staged by the library)
    for args, pinned in ((["0", "200", "6", "3"], 1), (["0", "1500", "3", "8"], 1), (["0", "333", "4", "2", "0"], 0)):
        out = subprocess.run([EXE] + args, capture_output=True, text=True)
        assert out.returncode == 0, out.stdout + out.stderr
        assert "samples_differ 0 status_differ 0" in out.stdout and "pinned %d" % pinned in out.stdout and "shards 2" in out.stdout, out.stdout
