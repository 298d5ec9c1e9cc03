#!/usr/bin/env python3
"""GPU box: what makes the first ~100 blocks of a FRESH C2 batch slower than its steady state -- the data (blanker average settling,
AGC attack) or the machine (clocks, TLB)?  All batches are created up front; the clocks are settled on a scratch batch; then every
variant runs its first launches, timed in groups of `G` (one HIP-event pair per group on the caller's stream), with a few scratch
launches right before so that no variant starts on an idle GPU.  Variants: the C2 settings, blanker off, AGC off, both off, and a
batch that has already processed 400 blocks (steady state, same machine conditions).
    python tools/transient_probe.py [groups] [G]"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

import audiosdr_amd as A
import bench

n_ch = 65536
NG = int(sys.argv[1]) if len(sys.argv) > 1 else 16
G = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
dI, dQ = bench.tiled_input(np, torch, dev, n_ch, 4, n_ch // 4, fc=6290.0, A=0.25)
dOut = torch.empty((n_ch, 128), dtype=torch.int16, device=dev)


def make(nb=True, agc=True):
    b = A.AudioSDRBatch(n_ch, device=0)
    bench.configure_c2(b)
    if not nb:
        b.disableNoiseBlanker()
    if not agc:
        b.disableAGC()
    return b


def run(b, n, first=0, s=stream):
    for i in range(first, first + n):
        b.update_device(dI[i % 4].data_ptr(), dQ[i % 4].data_ptr(), dOut.data_ptr(), 1, s)


variants = {"c2": make(), "nb_off": make(nb=False), "agc_off": make(agc=False), "both_off": make(False, False), "c2_settled": make()}
fresh_for_lanes = make()       # created up front as well: creating a batch idles the GPU and the clocks fall back
fresh_per_launch = make()
fresh_lanes2 = make()
fresh_lanes3 = make(False, False)
scratch = make()
run(scratch, 1500)
run(variants["c2_settled"], 400)
torch.cuda.synchronize()
out = {}
import time


def timed_groups(b, s):
    run(scratch, 200)
    torch.cuda.synchronize()
    run(scratch, 20)
    if s == A.STREAM_BATCH:   # lanes have no single stream to record on: host clock around a sync per group
        ts = []
        torch.cuda.synchronize()
        for g in range(NG):
            t0 = time.perf_counter()
            run(b, G, first=g * G, s=s)
            b.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3 / G)
        return [round(x, 4) for x in ts]
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(NG + 1)]
    ev[0].record()
    for g in range(NG):
        run(b, G, first=g * G)
        ev[g + 1].record()
    torch.cuda.synchronize()
    return [round(ev[g].elapsed_time(ev[g + 1]) / G, 4) for g in range(NG)]


for name, b in variants.items():
    out[name] = timed_groups(b, stream)
for name, b in (("settled", variants["c2_settled"]), ("fresh1", fresh_for_lanes), ("settled_again", variants["c2_settled"]),
                ("fresh2", fresh_lanes2), ("fresh_both_off", fresh_lanes3), ("settled_third", variants["c2_settled"])):
    out["lanes_" + name] = timed_groups(b, A.STREAM_BATCH)
run(scratch, 200)
torch.cuda.synchronize()
run(scratch, 20)
NL = 40
ev = [torch.cuda.Event(enable_timing=True) for _ in range(NL + 1)]
ev[0].record()
for i in range(NL):
    run(fresh_per_launch, 1, first=i)
    ev[i + 1].record()
torch.cuda.synchronize()
out["c2_per_launch"] = [round(ev[i].elapsed_time(ev[i + 1]), 4) for i in range(NL)]
print(json.dumps(out))
for k, v in out.items():
    print(f"{k:28s}", " ".join(f"{x:.4f}" for x in v), file=sys.stderr)
