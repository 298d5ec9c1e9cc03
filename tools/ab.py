#!/usr/bin/env python3
"""Same-process steady-state A/B of library builds on named workloads (GPU box).

  python tools/ab.py build name=-DFLAG1,-DFLAG2 ...     here: audiosdr_amd/variants/libasdr_<name>.so (name alone = no extra flags)
  python tools/ab.py run <workload>[,<workload>...] [rounds] [n_rep]      GPU box: every variant in audiosdr_amd/variants/ + the in-tree library

Workloads (65,536 channels unless stated; one block per launch; ms per launch from ONE event pair around n_rep launches):
  c2n2 / c2n3 / c2n4 / c2h   c2 at 131,072 / 196,608 / 262,144 / 73,728 (= 3 full rounds of 12 waves per CU) channels
  c2      bench.py's headline            c2div   8 different mixer phases per wave      c2agc   AGC hang time 0
  c2imp   an impulse in every block      c2adv   the three together                     c2x0    c2 without the kept audio row
  c3      SAM, 262,144 ch                c4      mode = c mod 7 + ALS, 131,072 ch       als1    all USB + ALS, 131,072 ch
  am      all AM, 131,072 ch             c4big   c4 at 1,048,576 ch             c4s / c4bigs / als1s   the same with the ALS filter as a launch of its own
Variants are timed interleaved, `rounds` times each; medians are printed, with the ratio to the in-tree library ("tree").
"""
import ctypes as C
import glob
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
VDIR = os.path.join(ROOT, "audiosdr_amd", "variants")
# AB_STREAM=batch: the calls go to ASDR_STREAM_BATCH (the batch's own streams: lanes) instead of the null stream
STREAM = C.c_void_p((1 << 64) - 1) if os.environ.get("AB_STREAM") == "batch" else None


def build(specs):
    from audiosdr_amd import build as b
    os.makedirs(VDIR, exist_ok=True)
    for spec in specs:
        name, _, flags = spec.partition("=")
        out = os.path.join(VDIR, "libasdr_%s.so" % name)
        b.build(force=True, extra_flags=[f for f in flags.split(",") if f], out=out)
        print("built", out, flags)


def setup(L, h, wl, n_ch, step):
    vp = C.c_void_p
    if wl in ("c2", "c2x0", "c2div", "c2agc", "c2imp", "c2adv", "c2n2", "c2n3", "c2n4", "c2h"):
        if wl in ("c2div", "c2adv"):
            L.asdr_setDemodMode(h, -1, 0)
            for j in range(8):
                for c in range(j, n_ch, 8):
                    L.asdr_setDemodMode(h, c, 1)
                step(j)
        L.asdr_setDemodMode(h, -1, 1); L.asdr_enableAudioFilter(h, -1)
        if wl in ("c2agc", "c2adv"):
            L.asdr_setAGChangTime(h, -1, 0.0)
        if wl == "c2x0" and hasattr(L, "asdr_set_exact_unknown_mode"):
            L.asdr_set_exact_unknown_mode(h, 0)
    elif wl == "c3":
        L.asdr_setDemodMode(h, -1, 5); L.asdr_setNoiseBlankerThresholdDb(h, -1, 10.0); L.asdr_enableAudioFilter(h, -1); L.asdr_setAudioFilter(h, -1, 0)
    elif wl in ("c4", "c4big", "c4s", "c4bigs"):
        for c in range(n_ch):
            L.asdr_setDemodMode(h, c, c % 7)
        L.asdr_enableALSfilter(h, -1); L.asdr_setNoiseBlankerThresholdDb(h, -1, 10.0)
        if wl.endswith("s"):
            L.asdr_set_als_launch_form(h, 1)   # short ALS filters as a launch of their own
    elif wl in ("als1", "als1s"):
        L.asdr_setDemodMode(h, -1, 1); L.asdr_enableALSfilter(h, -1); L.asdr_setNoiseBlankerThresholdDb(h, -1, 10.0)
        if wl.endswith("s"):
            L.asdr_set_als_launch_form(h, 1)
    elif wl == "am":
        L.asdr_setDemodMode(h, -1, 4); L.asdr_setNoiseBlankerThresholdDb(h, -1, 10.0)
    else:
        raise SystemExit("unknown workload " + wl)


def signal(wl):
    if wl == "c3":
        return dict(fc=None, A=0.3, m=0.5, fm=400.0)
    if wl in ("c4", "c4big", "als1", "am", "c4s", "c4bigs", "als1s"):
        return dict(fc=6890.0 - 300, A=0.3, m=0.4, f2=7500.0, a2=0.15)
    s = dict(fc=6290.0, A=0.25)
    if wl in ("c2imp", "c2adv"):
        s.update(impulse_every=128)
    return s


def run(workloads, rounds, n_rep):
    import audiosdr_amd as A
    import bench
    import torch
    paths = {"tree": A.library_path()}
    for f in sorted(glob.glob(os.path.join(VDIR, "libasdr_*.so"))):
        paths[os.path.basename(f)[8:-3]] = f
    only = os.environ.get("AB_ONLY")
    if only:
        paths = {k: v for k, v in paths.items() if k in only.split(",") or k == "tree"}
    dev = torch.device("cuda", 0)
    for wl in workloads:
        n_ch = {"c3": 262144, "c4": 131072, "als1": 131072, "am": 131072, "c4big": 1048576, "c4s": 131072, "als1s": 131072, "c4bigs": 1048576, "c2n2": 131072, "c2n3": 196608, "c2n4": 262144,
                "c2h": 73728}.get(wl, 65536)
        sig = signal(wl)
        uniq = 3584 if wl in ("c3", "c4", "c4big", "als1", "am", "c4s", "c4bigs", "als1s") else n_ch // 4
        if sig.get("fc") is None:
            sig["fc"] = 6890.0 + (np.arange(uniq) % 7 - 3) * 50.0
        dI, dQ = bench.tiled_input(np, torch, dev, n_ch, 4, uniq, **sig)
        dOut = torch.empty((n_ch, 128), dtype=torch.int16, device=dev)
        libs = {}
        for name, p in paths.items():
            L = A.binding.load_library(p)
            h = L.asdr_create(n_ch, 0)

            def step(i, L=L, h=h):
                L.asdr_update_device(h, C.c_void_p(dI[i & 3].data_ptr()), C.c_void_p(dQ[i & 3].data_ptr()), C.c_void_p(dOut.data_ptr()), 1, STREAM)
            setup(L, h, wl, n_ch, step)
            libs[name] = (L, h, step)
        times = {k: [] for k in libs}
        for r in range(rounds + 1):
            for name, (L, h, step) in libs.items():
                for i in range(max(40, n_rep // 4)):
                    step(i)
                L.asdr_region_timing_begin(h, STREAM)
                for i in range(n_rep):
                    step(i)
                total, calls = C.c_float(0.0), C.c_long(0)
                L.asdr_region_timing_end(h, C.byref(total), C.byref(calls))
                if r > 0:
                    times[name].append(total.value / max(1, calls.value))
        base = float(np.median(times["tree"]))
        for k, v in times.items():
            print("%-7s %-14s median %.5f ms  min %.5f  x%.4f of tree" % (wl, k, float(np.median(v)), float(np.min(v)), float(np.median(v)) / base), flush=True)
        for name, (L, h, step) in libs.items():
            L.asdr_destroy(h)
        del dI, dQ, dOut
        torch.cuda.empty_cache()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        build(sys.argv[2:])
    else:
        wls = sys.argv[2].split(",") if len(sys.argv) > 2 else ["c2"]
        run(wls, int(sys.argv[3]) if len(sys.argv) > 3 else 5, int(sys.argv[4]) if len(sys.argv) > 4 else 600)
