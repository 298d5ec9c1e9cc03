#!/usr/bin/env python3
"""GPU box: the C2 launch (65,536 channels x 1 block, bench.py's settings and input) as 1 .. 8 kernels on as many streams
(asdr_set_launch_split), and as independent shards that never join (asdr_create_sharded on one device, every shard on its own
stream): milliseconds per step, one HIP-event pair / host clock around 600 back-to-back steps, interleaved rounds, medians.
    python tools/split_probe.py [channels] [rounds]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

import audiosdr_amd as A
import bench

n_ch = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
dI, dQ = bench.tiled_input(np, torch, dev, n_ch, 4, n_ch // 4, fc=6290.0, A=0.25)
dOut = torch.empty((n_ch, 128), dtype=torch.int16, device=dev)
cases = {}
for sp in (1, 2, 3, 4, 6, 8):
    b = A.AudioSDRBatch(n_ch, device=0)
    bench.configure_c2(b)
    b.set_launch_split(sp, 256)
    cases["split%d" % sp] = b


def run(b, steps):
    for i in range(steps):
        b.update_device(dI[i % 4].data_ptr(), dQ[i % 4].data_ptr(), dOut.data_ptr(), 1, stream)


res = {k: [] for k in cases}
for b in cases.values():
    run(b, 600)
torch.cuda.synchronize()
for r in range(rounds):
    for k, b in cases.items():
        run(b, 50)
        torch.cuda.synchronize()
        b.region_timing_begin(stream)
        run(b, 600)
        ms, calls = b.region_timing_end()
        res[k].append(ms / calls)
out = {k: round(float(np.median(v)), 5) for k, v in res.items()}
for b in cases.values():
    b.close()
# independent shards on one device, each on its own stream, never joined
for G in (2, 3, 4):
    sb = A.AudioSDRBatch(n_ch, devices=[0] * G)
    bench.configure_c2(sb)
    views, strs = [sb.shard(g) for g in range(G)], [torch.cuda.Stream(device=dev) for _ in range(G)]
    rng = [sb.shard_range(g) for g in range(G)]

    def step(i):
        for v, st, (lo, hi) in zip(views, strs, rng):
            v.update_device(dI[i % 4].data_ptr() + lo * 256, dQ[i % 4].data_ptr() + lo * 256, dOut.data_ptr() + lo * 256, 1, st.cuda_stream)
    for i in range(600):
        step(i)
    torch.cuda.synchronize()
    ts = []
    for r in range(rounds):
        t0 = time.perf_counter()
        for i in range(600):
            step(i)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 600 * 1e3)
    out["shards%d_unjoined" % G] = round(float(np.median(ts)), 5)
    for v in views:
        v.close()
    sb.close()
print(json.dumps({"channels": n_ch, "ms_per_step": out}))
