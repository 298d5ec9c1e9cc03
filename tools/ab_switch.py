#!/usr/bin/env python3
"""Same-process A/B of a batch-level switch on the C2 workload (GPU box): steady-state ms per launch with
asdr_set_exact_unknown_mode on / off (the post-ALS row every block stores so that unknown mode values can be re-processed as
the reference does, +512 B written per channel-block).  One HIP-event pair around each run of `n_rep` back-to-back launches,
runs interleaved.    python tools/ab_switch.py [rounds] [n_rep]"""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import audiosdr_amd as A  # noqa: E402
import bench  # noqa: E402
import torch  # noqa: E402


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 7
    n_rep = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    n_ch = 65536
    dev = torch.device("cuda", 0)
    dI, dQ = bench.tiled_input(np, torch, dev, n_ch, 4, n_ch // 4, fc=6290.0, A=0.25)
    dOut = torch.empty((n_ch, 128), dtype=torch.int16, device=dev)
    batches = {}
    for name, on in (("exact_on", True), ("exact_off", False)):
        b = A.AudioSDRBatch(n_ch)
        bench.configure_c2(b)
        b.set_exact_unknown_mode(on)
        batches[name] = b
    times = {k: [] for k in batches}
    for r in range(rounds + 1):
        for name, b in batches.items():
            def step(i, b=b):
                b.update_device(dI[i & 3].data_ptr(), dQ[i & 3].data_ptr(), dOut.data_ptr(), 1, 0)
            ms = bench.measure_region(b, step, 0, n_rep // 4, n_rep)
            if r > 0:
                times[name].append(ms)
    for k, v in times.items():
        print("%-10s steady median %.5f ms  min %.5f ms  (%d runs of %d launches)" % (k, float(np.median(v)), float(np.min(v)), len(v), n_rep))
    print("exact_on / exact_off = %.4f" % (float(np.median(times["exact_on"])) / float(np.median(times["exact_off"]))))


if __name__ == "__main__":
    main()
