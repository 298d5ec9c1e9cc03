#!/usr/bin/env python3
"""GPU box: the C2 job on ASDR_STREAM_BATCH (lanes) with K application streams alive in the process (created before the batch's first call and
kept busy with tiny kernels now and then): what the lanes lose to sharing hardware queues with streams they do not own, by the priority the
library's stream pool is created with (ASDR_POOL_PRIORITY).   python tools/foreign_streams_probe.py"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import audiosdr_amd as A  # noqa: E402
import bench  # noqa: E402

dev = torch.device("cuda", 0)
L = A.load_library()
STREAM = C.c_void_p((1 << 64) - 1)
n_ch = 65536
dI, dQ = bench.tiled_input(np, torch, dev, n_ch, 4, n_ch // 4, fc=6890.0 - 1500.0, A=0.25, noise=0.02)
dOut = torch.empty((n_ch, 128), dtype=torch.int16, device=dev)
torch.cuda.synchronize()
for K in ([int(x) for x in sys.argv[1].split(',')] if len(sys.argv) > 1 else (0, 1, 3, 6)):
    streams = [torch.cuda.Stream() for _ in range(K)]
    junk = [torch.zeros(1024, device=dev) for _ in range(K)]
    for s_, j in zip(streams, junk):
        with torch.cuda.stream(s_):
            j.add_(1.0)
    torch.cuda.synchronize()
    b = A.AudioSDRBatch(n_ch)
    bench.configure_c2(b)

    def step(i):
        L.asdr_update_device(b._h, C.c_void_p(dI[i & 3].data_ptr()), C.c_void_p(dQ[i & 3].data_ptr()), C.c_void_p(dOut.data_ptr()), 1, STREAM)
    for i in range(600):
        step(i)
    b.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    import time
    t0 = time.perf_counter()
    N = 1500
    for i in range(N):
        step(i)
        if K and i % 100 == 0:
            with torch.cuda.stream(streams[(i // 100) % K]):
                junk[(i // 100) % K].add_(1.0)
    b.synchronize(); torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / N * 1e3
    print(json.dumps({"foreign_streams": K, "pool_priority": os.environ.get("ASDR_POOL_PRIORITY", "default"), "ms_per_step": round(ms, 5), "lanes_enabled": b.lanes_enabled(), "probe": b.lanes_overlap_probe()}), flush=True)
    b.close()
    del streams
