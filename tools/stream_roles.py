#!/usr/bin/env python3
"""Which role bounds the streaming block pipeline?  C5-shaped calls (512 WSPR channels x 646 blocks) timed for library variants
with one stage compiled out (tools/ablate.py build: audiosdr_amd/variants/libasdr_{full,no_IF,no_MIX,no_HIL,no_AGC}.so; ablated
outputs are wrong by construction, only the time matters).  (GPU box.)   python tools/stream_roles.py [blocks]"""
import ctypes as C
import glob
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
VDIR = os.path.join(ROOT, "audiosdr_amd", "variants")


def main():
    import torch
    import audiosdr_amd as A
    from audiosdr_amd.synth import make_iq
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 646
    n_ch = 512
    I, Q = make_iq(n_ch, T, fc=6890.0 - 1500.0, A=0.02, noise=0.05)
    dI = torch.from_numpy(I).cuda(); dQ = torch.from_numpy(Q).cuda()
    dO = torch.empty((n_ch, T, 128), dtype=torch.int16, device="cuda")
    names = os.environ.get("STREAM_ROLES_ONLY", "full,no_IF,no_MIX,no_HIL,no_AGC").split(",")
    for name in names:
        p = os.path.join(VDIR, "libasdr_%s.so" % name)
        if not os.path.exists(p):
            continue
        L = A.binding.load_library(p)
        h = L.asdr_create(n_ch, 0)
        L.asdr_disableNoiseBlanker(h, -1); L.asdr_setAGCmode(h, -1, 2); L.asdr_setDemodMode(h, -1, 6)
        ms = []
        for rep in range(6):
            L.asdr_region_timing_begin(h, None)
            L.asdr_update_device(h, C.c_void_p(dI.data_ptr()), C.c_void_p(dQ.data_ptr()), C.c_void_p(dO.data_ptr()), T, None)
            t, n = C.c_float(0), C.c_long(0)
            L.asdr_region_timing_end(h, C.byref(t), C.byref(n))
            ms.append(t.value)
        print("%-10s %8.3f ms per %d-block call = %6.2f us per block   (pipeline launches: %d)" % (
            name, float(np.median(ms[1:])), T, float(np.median(ms[1:])) * 1e3 / T, L.asdr_stream_pipeline_launches(h)))
        L.asdr_destroy(h)


if __name__ == "__main__":
    main()
