#!/usr/bin/env python3
"""Phase ablation of asdr_update_kernel: builds variants with -DASDR_ABLATE=<mask> and times them
interleaved in ONE process on the C2 workload (cdna_hip_programming.md 5.4 rule 24).  Outputs of ablated builds
are wrong by construction; only their kernel time matters.

  python tools/ablate.py build          # here (hipcc cross-compiles): audiosdr_amd/variants/*.so
  python tools/ablate.py run [rounds]   # on the GPU box
"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
VDIR = os.path.join(ROOT, "audiosdr_amd", "variants")
PHASES = {"NB": 1, "IF": 2, "SAM": 4, "MIX": 8, "HIL": 16, "ENV": 32, "AF": 64, "AGC": 128, "ALS": 256}
VARIANTS = {"full": 0, "no_NB": 1, "no_IF": 2, "no_MIX": 8, "no_HIL": 16, "no_AF": 64, "no_AGC": 128,
            "io_only": 511}
EXTRA = {"full_w2": ["-DASDR_WAVES_PER_EU=2"],       # same code, 256-VGPR budget (8 waves/CU)
         "chunk4": ["-DASDR_PIPE_CHUNK=4"],          # biquad pipeline with 4-sample chunks (35 steps)
         "nb_general": ["-DASDR_NB_ALWAYS_SLOW"]}    # blanker always on its general path (no quiet fast path)
# Any other libasdr_<name>.so dropped into audiosdr_amd/variants/ (e.g. a build of an older commit) is timed as well.


def build():
    from audiosdr_amd import build as b
    os.makedirs(VDIR, exist_ok=True)
    for name, mask in VARIANTS.items():
        out = os.path.join(VDIR, "libasdr_%s.so" % name)
        b.build(force=True, extra_flags=["-DASDR_ABLATE=%d" % mask], out=out)
        print("built", out)
    for name, flags in EXTRA.items():
        out = os.path.join(VDIR, "libasdr_%s.so" % name)
        b.build(force=True, extra_flags=flags, out=out)
        print("built", out)


def run(rounds=5, n_ch=65536):
    import audiosdr_amd as A
    from audiosdr_amd.synth import make_iq
    import torch
    uniq = min(4096, n_ch)
    I, Q = make_iq(uniq, 2, fc=6290.0, A=0.25)
    I = np.tile(I, (n_ch // uniq, 1, 1)); Q = np.tile(Q, (n_ch // uniq, 1, 1))
    dI = [torch.from_numpy(np.ascontiguousarray(I[:, b])).cuda() for b in range(2)]
    dQ = [torch.from_numpy(np.ascontiguousarray(Q[:, b])).cuda() for b in range(2)]
    dOut = torch.empty((n_ch, 128), dtype=torch.int16, device="cuda")
    libs = {}
    import glob
    names = list(VARIANTS) + list(EXTRA)
    names += [os.path.basename(f)[8:-3] for f in sorted(glob.glob(os.path.join(VDIR, "libasdr_*.so"))) if os.path.basename(f)[8:-3] not in names]
    only = os.environ.get("ABLATE_ONLY")
    if only:
        names = [n for n in names if n in only.split(",")]
    for name in names:
        p = os.path.join(VDIR, "libasdr_%s.so" % name)
        if not os.path.exists(p):
            continue
        L = A.binding.load_library(p)
        h = L.asdr_create(n_ch, 0)
        if hasattr(L, "asdr_set_launch_timing"):
            L.asdr_set_launch_timing(h, 1)   # an event pair per call (older variant builds always record it)
        cfg = os.environ.get("ABLATE_CFG", "c2")
        if cfg == "c3":      # SAM + PLL (BASELINE config 3 settings)
            L.asdr_setDemodMode(h, -1, 5); L.asdr_setNoiseBlankerThresholdDb(h, -1, 10.0); L.asdr_enableAudioFilter(h, -1); L.asdr_setAudioFilter(h, -1, 0)
        elif cfg == "c4":    # mixed modes + ALS (BASELINE config 4 settings)
            for c in range(n_ch):
                L.asdr_setDemodMode(h, c, c % 7)
            L.asdr_enableALSfilter(h, -1); L.asdr_setNoiseBlankerThresholdDb(h, -1, 10.0)
        elif cfg == "c4blk":   # the C4 mix with the seven modes in contiguous channel ranges instead of c mod 7
            for c in range(n_ch):
                L.asdr_setDemodMode(h, c, min(6, c * 7 // n_ch))
            L.asdr_enableALSfilter(h, -1); L.asdr_setNoiseBlankerThresholdDb(h, -1, 10.0)
        elif cfg == "c4noals":
            for c in range(n_ch):
                L.asdr_setDemodMode(h, c, c % 7)
            L.asdr_setNoiseBlankerThresholdDb(h, -1, 10.0)
        elif cfg.startswith("als_mode"):   # one mode + ALS: where a C4 regression comes from
            L.asdr_setDemodMode(h, -1, int(cfg[8:])); L.asdr_enableALSfilter(h, -1); L.asdr_setNoiseBlankerThresholdDb(h, -1, 10.0)
        elif cfg.startswith("mode"):       # one mode, C4 blanker setting, no ALS
            L.asdr_setDemodMode(h, -1, int(cfg[4:])); L.asdr_setNoiseBlankerThresholdDb(h, -1, 10.0)
        elif cfg == "c5":    # WSPR receiver settings, blanker and audio filter off
            L.asdr_disableNoiseBlanker(h, -1); L.asdr_setAGCmode(h, -1, 2); L.asdr_setDemodMode(h, -1, 6)
        else:
            L.asdr_setDemodMode(h, -1, 1)
            L.asdr_enableAudioFilter(h, -1)
        libs[name] = (L, h)
    times = {k: [] for k in libs}
    VARS = libs
    if os.environ.get("ABLATE_STEADY"):
        # steady state instead of a HIP-event pair per call: back-to-back launches on the resident buffers, ONE event pair around
        # each run of `n_rep` launches (as bench.py measures) -- resolves differences of 1 % that the per-call figures do not
        n_rep = int(os.environ.get("ABLATE_STEADY"))
        n_rep = n_rep if n_rep > 1 else 1000
        libs = {k: v for k, v in libs.items() if hasattr(v[0], "asdr_region_timing_begin")}   # (older variants lack the API)
        times = {k: [] for k in libs}
        for name, (L, h) in libs.items():
            L.asdr_set_launch_timing(h, 0)
        for r in range(rounds + 1):
            for name, (L, h) in libs.items():
                def go(n):
                    for i in range(n):
                        L.asdr_update_device(h, C.c_void_p(dI[i & 1].data_ptr()), C.c_void_p(dQ[i & 1].data_ptr()), C.c_void_p(dOut.data_ptr()), 1, None)
                go(n_rep // 4)
                L.asdr_region_timing_begin(h, None)
                go(n_rep)
                total, calls = C.c_float(0.0), C.c_long(0)
                L.asdr_region_timing_end(h, C.byref(total), C.byref(calls))
                if r > 0:
                    times[name].append(total.value / max(1, calls.value))
        for k, v in times.items():
            print("%-12s steady median %.4f ms  min %.4f ms  (%d runs of %d launches)" % (k, float(np.median(v)), float(np.min(v)), len(v), n_rep))
        return
    for r in range(rounds + 1):
        for name, (L, h) in libs.items():
            for i in range(6):
                L.asdr_update_device(h, C.c_void_p(dI[i & 1].data_ptr()), C.c_void_p(dQ[i & 1].data_ptr()),
                                     C.c_void_p(dOut.data_ptr()), 1, None)
                ms = L.asdr_last_kernel_ms(h)
                if r > 0 and i >= 2:
                    times[name].append(ms)
    res = {k: {"median_ms": float(np.median(v)), "min_ms": float(np.min(v))} for k, v in times.items()}
    full = res.get("full", {}).get("median_ms")
    for k, v in res.items():
        v["delta_vs_full_ms"] = None if full is None else round(full - v["median_ms"], 4)
        print("%-10s median %.4f ms  min %.4f ms  (full - this = %s ms)" % (k, v["median_ms"], v["min_ms"], v["delta_vs_full_ms"]))
    print(json.dumps(res))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        build()
    else:
        # ABLATE_NCH=65536,2048,8192,16384: also time smaller batches (1 wave/CU, 1 and 2 waves/SIMD) -> lone-wave latency
        for n in [int(x) for x in os.environ.get("ABLATE_NCH", "65536").split(",")]:
            print("== %d channels" % n)
            run(int(sys.argv[2]) if len(sys.argv) > 2 else 5, n)
