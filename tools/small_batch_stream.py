#!/usr/bin/env python3
"""GPU box: what a SMALL receiver bank costs per block in a T-block streaming call, by settings family -- the block pipeline's role sets
(WSPR / SSB, AM), and the families that have none and run the in-kernel block loop (audio filter on, SAM, ALS): the floor each of them
has is its longest per-channel dependent chain.  One line per family and bank size.
    python tools/small_batch_stream.py [T]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import audiosdr_amd as A  # noqa: E402
import bench  # noqa: E402
from audiosdr_amd.synth import make_iq  # noqa: E402
import torch  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 128
FAMILIES = {
    "wspr (BareBonesWSPR settings)": lambda s: bench.configure_c5(s),
    "wspr + audio filter (C5 variant B)": lambda s: (bench.configure_c5(s), s.enableAudioFilter()),
    "usb + blanker + audio filter + agc (C2 settings)": lambda s: bench.configure_c2(s),
    "am + audio filter": lambda s: (s.setDemodMode(4), s.enableAudioFilter(), s.setNoiseBlankerThresholdDb(10.0)),
    "sam + audio filter (C3 settings)": lambda s: (s.setDemodMode(5), s.setNoiseBlankerThresholdDb(10.0), s.enableAudioFilter(), s.setAudioFilter(0)),
    "usb + als": lambda s: (s.setDemodMode(1), s.enableALSfilter(), s.setNoiseBlankerThresholdDb(10.0)),
}
for n_ch in (512, 4096):
    I, Q = make_iq(n_ch, T, fc=6890.0 - 300, A=0.3, m=0.4, noise=0.01)
    dI, dQ = torch.from_numpy(I).cuda(), torch.from_numpy(Q).cuda()
    dO = torch.empty((n_ch, T, 128), dtype=torch.int16, device="cuda")
    for name, cfg in FAMILIES.items():
        b = A.AudioSDRBatch(n_ch)
        cfg(b)
        for _ in range(2):
            b.update_device(dI.data_ptr(), dQ.data_ptr(), dO.data_ptr(), T, 0)
        b.synchronize()
        p0 = b.stream_pipeline_launches(); r0 = b.als_role_calls()
        t0 = time.perf_counter()
        for _ in range(4):
            b.update_device(dI.data_ptr(), dQ.data_ptr(), dO.data_ptr(), T, 0)
        b.synchronize()
        ms = (time.perf_counter() - t0) / 4 * 1e3
        print(json.dumps({"family": name, "channels": n_ch, "T": T, "ms_per_call": round(ms, 3), "us_per_block": round(ms * 1e3 / T, 2),
                          "times_real_time": round(T * 128 / 44100.0 / (ms * 1e-3), 1), "pipeline_calls": b.stream_pipeline_launches() - p0, "als_role_calls": b.als_role_calls() - r0}), flush=True)
        b.close()
