#!/usr/bin/env python3
"""GPU box: the C2 step (lanes off) back to back on the null stream, on a stream torch creates, on a hipStreamCreate'd blocking stream,
on a non-blocking one, and on ASDR_STREAM_BATCH (the batch's pool stream): ms per step, one event pair around 400 steps."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

import audiosdr_amd as A
import bench

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from helpers import Hip

n_ch = 65536
dev = torch.device("cuda", 0)
dI, dQ = bench.tiled_input(np, torch, dev, n_ch, 4, n_ch // 4, fc=6290.0, A=0.25)
dOut = torch.empty((n_ch, 128), dtype=torch.int16, device=dev)
hip = Hip()
nb = C.c_void_p()
hip.h.hipStreamCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
assert hip.h.hipStreamCreateWithFlags(C.byref(nb), 1) == 0
streams = {"null": 0, "torch_stream": torch.cuda.Stream(device=dev).cuda_stream, "hip_blocking": hip.stream(), "hip_nonblocking": nb.value,
           "batch": A.STREAM_BATCH}
out = {}
for lanes in (0, 2):
    for name, s in streams.items():
        if lanes and name != "batch":
            continue
        b = A.AudioSDRBatch(n_ch, device=0)
        bench.configure_c2(b)
        b.set_lanes(lanes)
        ts = []
        for r in range(4):
            for i in range(100):
                b.update_device(dI[i % 4].data_ptr(), dQ[i % 4].data_ptr(), dOut.data_ptr(), 1, s)
            b.synchronize()
            b.region_timing_begin(s)
            for i in range(400):
                b.update_device(dI[i % 4].data_ptr(), dQ[i % 4].data_ptr(), dOut.data_ptr(), 1, s)
            ms, calls = b.region_timing_end()
            ts.append(ms / calls)
        out["%s_lanes%d" % (name, lanes)] = round(float(np.median(ts)), 5)
        b.close()
print(json.dumps(out))
