#!/usr/bin/env python3
"""GPU box: per-launch kernel time (one HIP-event pair per launch) of the first launches of a FRESH C2 batch, right after 1500
launches of a scratch batch (clocks settled) -- what a short run (the driver's --steps 20 --warmup 5) measures.
    python tools/warmup_curve.py [launches]"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

import audiosdr_amd as A
import bench

n_ch = 65536
N = int(sys.argv[1]) if len(sys.argv) > 1 else 120
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
dI, dQ = bench.tiled_input(np, torch, dev, n_ch, 4, n_ch // 4, fc=6290.0, A=0.25)
dOut = torch.empty((n_ch, 128), dtype=torch.int16, device=dev)


def fresh():
    b = A.AudioSDRBatch(n_ch, device=0)
    bench.configure_c2(b)
    return b


s = fresh()
for i in range(1500):
    s.update_device(dI[i % 4].data_ptr(), dQ[i % 4].data_ptr(), dOut.data_ptr(), 1, stream)
torch.cuda.synchronize()
s.close()
out = {}
for name in ("fresh_batch", "second_fresh_batch"):
    b = fresh()
    b.kernel_timing_begin(N)
    for i in range(N):
        b.update_device(dI[i % 4].data_ptr(), dQ[i % 4].data_ptr(), dOut.data_ptr(), 1, stream)
    ms = b.kernel_timing_end(N)
    out[name] = [round(float(x), 4) for x in ms]
    b.close()
print(json.dumps(out))
