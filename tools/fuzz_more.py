#!/usr/bin/env python3
"""One-off deep fuzz on a GPU box: the suite's randomised parity tests (tests/test_gpu_fuzz.py) with seeds the suite does not use.
    python tools/fuzz_more.py [first_seed] [n_seeds] [seconds]      (stops after `seconds`; prints one line per test and seed)"""
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401,E402  (one HIP runtime per process: INTEGRATION.md 5)
import audiosdr_amd as gpu  # noqa: E402
from oracle import asdr_oracle as ao  # noqa: E402
import test_gpu_fuzz as F  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
budget = float(sys.argv[3]) if len(sys.argv) > 3 else 240.0
ao.build(); ao.lib()
ao.PRODUCT_PLL_BOUND = True
t0 = time.time()
bad = 0
for seed in range(first, first + n):
    for name, args in (("test_fuzz_control_surface", (seed,)), ("test_fuzz_whole_waves", (seed,)), ("test_fuzz_large_mixed_batch", (seed,)),
                       ("test_fuzz_multi_block_calls", (seed, False)), ("test_fuzz_multi_block_calls", (seed, True))):
        if time.time() - t0 > budget:
            print("time budget reached; failures:", bad); sys.exit(1 if bad else 0)
        try:
            getattr(F, name)(gpu, ao, *args)
            print("ok  ", name, args, flush=True)
        except AssertionError as e:
            bad += 1
            print("FAIL", name, args, str(e)[:300], flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
