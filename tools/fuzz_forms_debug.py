#!/usr/bin/env python3
"""Replay one seed of tests/test_gpu_fuzz_launch_forms.py, one line per call and setter; with `sync` a synchronisation (and a tap
comparison) after EVERY call (GPU box):
    python tools/fuzz_forms_debug.py <seed> [sync]
A seed that fails in the suite and passes here with `sync` is an ordering bug between two launch forms (round 4, seed 155: a call the
block pipeline took had been classified as a lane call first and skipped its wait for the previous call on another stream)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: F401  (one HIP runtime per process: INTEGRATION.md 5)

import audiosdr_amd as gpu
import tests.test_gpu_fuzz_launch_forms as F

F.run_sequence(gpu, int(sys.argv[1]), sync_each=len(sys.argv) > 2 and sys.argv[2] == "sync", log=print)
print("seed", sys.argv[1], "ok")
