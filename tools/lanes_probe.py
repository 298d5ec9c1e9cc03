#!/usr/bin/env python3
"""GPU box: steady-state ms per step on ASDR_STREAM_BATCH for 0 (off) / 2 / 3 / 4 / 6 / 8 lanes (asdr_set_lanes), workloads of tools/ab.py.
    python tools/lanes_probe.py c2,c3,c4 [rounds] [n_rep]"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _variant  # noqa: E402,F401  (ASDR_TOOLS_LIB)
import numpy as np
import torch

import audiosdr_amd as A
import bench
import ab

wls = sys.argv[1].split(",") if len(sys.argv) > 1 else ["c2"]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
n_rep = int(sys.argv[3]) if len(sys.argv) > 3 else 400
dev = torch.device("cuda", 0)
L = A.load_library()
STREAM = C.c_void_p((1 << 64) - 1)
for wl in wls:
    n_ch = {"c3": 262144, "c4": 131072, "als1": 131072, "am": 131072, "c4big": 1048576}.get(wl, 65536)
    sig = ab.signal(wl)
    uniq = 3584 if wl in ("c3", "c4", "c4big", "als1", "am") else n_ch // 4
    if sig.get("fc") is None:
        sig["fc"] = 6890.0 + (np.arange(uniq) % 7 - 3) * 50.0
    dI, dQ = bench.tiled_input(np, torch, dev, n_ch, 4, uniq, **sig)
    dOut = torch.empty((n_ch, 128), dtype=torch.int16, device=dev)
    cases = {}
    for nl in (-1, 0, 2, 3, 4, 6, 8):   # -1: the library's default for the schedule
        h = L.asdr_create(n_ch, 0)

        def step(i, h=h):
            L.asdr_update_device(h, C.c_void_p(dI[i & 3].data_ptr()), C.c_void_p(dQ[i & 3].data_ptr()), C.c_void_p(dOut.data_ptr()), 1, STREAM)
        ab.setup(L, h, wl, n_ch, step)
        if nl >= 0:
            L.asdr_set_lanes(h, nl, 0)
        cases[nl] = (h, step)
    times = {k: [] for k in cases}
    for r in range(rounds + 1):
        for nl, (h, step) in cases.items():
            for i in range(max(40, n_rep // 4)):
                step(i)
            L.asdr_region_timing_begin(h, STREAM)
            for i in range(n_rep):
                step(i)
            total, calls = C.c_float(0.0), C.c_long(0)
            L.asdr_region_timing_end(h, C.byref(total), C.byref(calls))
            if r > 0:
                times[nl].append(total.value / max(1, calls.value))
    print(json.dumps({"workload": wl, "channels": n_ch, "ms_per_step_by_lanes": {("default" if k < 0 else str(k)): round(float(np.median(v)), 5) for k, v in times.items()}}), flush=True)
    for h, _ in cases.values():
        L.asdr_destroy(h)
    del dI, dQ, dOut
    torch.cuda.empty_cache()
