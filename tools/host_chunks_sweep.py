#!/usr/bin/env python3
"""GPU box: asdr_update() (host pointers, pinned caller buffers) on the C2 job by number of chunks (asdr_set_host_chunks): the H2D | kernels |
D2H overlap against the per-chunk cost of the events that chain the three streams.   python tools/host_chunks_sweep.py"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import audiosdr_amd as A  # noqa: E402
import bench  # noqa: E402
from audiosdr_amd.synth import make_iq  # noqa: E402

n_ch, NB = 65536, 4
uniq = n_ch // 16
bI, bQ = make_iq(uniq, NB, fc=6290.0, A=0.25)
reps = n_ch // uniq
pin = [A.host_alloc((n_ch, 1, 128)) for _ in range(2 * NB + 1)]
for b in range(NB):
    pin[2 * b][:] = np.tile(bI[:, b:b + 1], (reps, 1, 1)); pin[2 * b + 1][:] = np.tile(bQ[:, b:b + 1], (reps, 1, 1))
for chunks in (0, 1, 2, 3, 4, 6, 8, 12, 16):
    batch = A.AudioSDRBatch(n_ch)
    bench.configure_c2(batch)
    batch.set_host_chunks(chunks)
    for i in range(4):
        batch.update_into(pin[2 * (i % NB)], pin[2 * (i % NB) + 1], pin[-1])
    t0 = time.perf_counter()
    N = 40
    for i in range(N):
        batch.update_into(pin[2 * (i % NB)], pin[2 * (i % NB) + 1], pin[-1])
    dt = (time.perf_counter() - t0) / N
    print(json.dumps({"chunks_asked": chunks, "chunks": batch.host_path_info()["chunks"], "ms_per_call": round(dt * 1e3, 4), "pcie_GBps": round(768.0 * n_ch / dt / 1e9, 2)}), flush=True)
    batch.close()
