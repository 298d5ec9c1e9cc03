#!/usr/bin/env python3
"""GPU box: does a process that used asdr_host_autopin(1) on SMALL heap allocations (released correctly) abort in later, unrelated copies?
(Round 6: the GPU suite did, about every second run, at the first test behind tests/test_gpu_host_path.py.)   python tools/autopin_hazard.py [small|large|none] [rounds]"""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import audiosdr_amd as A  # noqa: E402
from helpers import Hip  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "small"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
n = {"small": 64, "large": 8192, "none": 64}[kind]
b = A.AudioSDRBatch(n)
b.setDemodMode(1)
if kind != "none":
    A.binding.host_autopin(1)
bufs = [[np.zeros((n, 1, 128), np.int16) for _ in range(3)] for _ in range(8)]
for rep in range(3):
    for tri in bufs:
        b.update_into(*tri)
print("registered", A.binding.host_autopin_info())
b.close()
A.binding.host_autopin_clear(); A.binding.host_autopin(0)
print("after clear", A.binding.host_autopin_info())
del bufs
hip = Hip()
rng = np.random.default_rng(1)
keep = []
for i in range(rounds):
    sz = int(rng.choice([4096, 16384, 65536, 262144, 1 << 20, 4 << 20]))
    a = rng.integers(-100, 100, sz // 2, dtype=np.int16)
    d = hip.upload(a)
    back = hip.download(d, a.shape, np.int16)
    assert np.array_equal(a, back)
    if i % 50 == 49:
        hip.free_all()
    if i % 7 == 0:
        keep.append(a)
    if len(keep) > 20:
        keep = keep[10:]
print("survived", rounds, "copies after", kind)
