#!/usr/bin/env python3
"""GPU box: ms per step by lane count measured the way no lane drift can flatter -- host clock from an idle GPU (asdr_synchronize) to a
drained one (asdr_synchronize) over N back-to-back calls on ASDR_STREAM_BATCH; beside it the library's region timing (HIP events on the
lanes, every lane held at the begin marker) over the same N calls.  Workloads as tools/ab.py (c2, c3, c4, als1, am, c4big).
    python tools/true_rate.py [workloads] [N] [lane counts, e.g. -1,0,2,3,4,6]        (-1 = the library's default for the schedule)"""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _variant  # noqa: E402,F401
import numpy as np
import torch

import audiosdr_amd as A
import bench
import ab

wls = (sys.argv[1] if len(sys.argv) > 1 else "c2,c3,c4").split(",")
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
counts = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "-1,0,2,3,4,6").split(",")]
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
    n = max(100, int(N * 65536 / n_ch)) if wl != "c2" else N
    out = {"workload": wl, "channels": n_ch, "calls": n, "host_clock_ms_per_step": {}, "region_ms_per_step": {}}
    for nl in counts:
        h = L.asdr_create(n_ch, 0)

        def step(i, h=h):
            L.asdr_update_device(h, C.c_void_p(dI[i & 3].data_ptr()), C.c_void_p(dQ[i & 3].data_ptr()), C.c_void_p(dOut.data_ptr()), 1, STREAM)
        ab.setup(L, h, wl, n_ch, step)
        if nl >= 0:
            L.asdr_set_lanes(h, nl, 0)
        for i in range(max(200, n // 4)):
            step(i)
        res = []
        for rep in range(3):
            L.asdr_synchronize(h)
            t0 = time.perf_counter()
            for i in range(n):
                step(i)
            L.asdr_synchronize(h)
            res.append((time.perf_counter() - t0) * 1e3 / n)
        reg = []
        for rep in range(3):
            for i in range(100):
                step(i)
            L.asdr_region_timing_begin(h, STREAM)
            for i in range(n):
                step(i)
            total, calls = C.c_float(0.0), C.c_long(0)
            L.asdr_region_timing_end(h, C.byref(total), C.byref(calls))
            reg.append(total.value / max(1, calls.value))
        key = "default" if nl < 0 else str(nl)
        out["host_clock_ms_per_step"][key] = round(float(np.median(res)), 5)
        out["region_ms_per_step"][key] = round(float(np.median(reg)), 5)
        L.asdr_destroy(h)
    print(json.dumps(out), flush=True)
    del dI, dQ, dOut
    torch.cuda.empty_cache()
