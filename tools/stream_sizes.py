#!/usr/bin/env python3
"""The block pipeline beyond one workgroup per compute unit (GPU box): WSPR-configured batches of n channels x T blocks per call,
time per call with the pipeline forced on (asdr_debug_set_stream_max_groups) against the in-kernel block loop, recoveries counted,
a sampled channel checked against the oracle.      python tools/stream_sizes.py [T] [am]
`am`: AM receivers (blanker at 10 dB, audio filter, AGC) instead -- the pipeline's AM role set."""
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import audiosdr_amd as A  # noqa: E402
import bench  # noqa: E402
from audiosdr_amd.synth import make_iq  # noqa: E402
from oracle import asdr_oracle as ao  # noqa: E402
import torch  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 256
AM = "am" in sys.argv[2:]


def configure(s):
    if AM:
        s.setDemodMode(4); s.enableAudioFilter(); s.setNoiseBlankerThresholdDb(10.0)
    else:
        bench.configure_c5(s)


for n_ch in ((64, 512, 1024, 4096) if AM else (512, 672, 680, 1024, 1344, 2048, 4096)):
    I, Q = make_iq(n_ch, T, fc=6890.0, A=0.3 if AM else 0.02, m=0.5 if AM else 0.0, noise=0.05)
    dI, dQ = torch.from_numpy(I).cuda(), torch.from_numpy(Q).cuda()
    dO = torch.empty((n_ch, T, 128), dtype=torch.int16, device="cuda")
    row = {"channels": n_ch, "groups": (n_ch + 7) // 8, "T": T}
    for mode in ("loop", "pipeline"):
        b = A.AudioSDRBatch(n_ch)
        configure(b)
        if mode == "loop":
            b.set_stream_pipeline(False)
        else:
            b.debug_set_stream_max_groups(1024)
        for _ in range(2):
            b.update_device(dI.data_ptr(), dQ.data_ptr(), dO.data_ptr(), T, 0)
        b.synchronize()
        t0 = time.perf_counter()
        for _ in range(4):
            b.update_device(dI.data_ptr(), dQ.data_ptr(), dO.data_ptr(), T, 0)
        b.synchronize()
        ms = (time.perf_counter() - t0) / 4 * 1e3
        got = dO[n_ch - 1].cpu().numpy()
        o = ao.OracleSDR(); configure(o)
        want = o.update(np.tile(I[n_ch - 1], (6, 1)), np.tile(Q[n_ch - 1], (6, 1))).reshape(6, T, 128)[5]
        row[mode] = {"ms_per_call": round(ms, 3), "us_per_block": round(ms * 1e3 / T, 2), "pipeline_calls": b.stream_pipeline_launches(),
                     "recoveries": b.stream_pipeline_recoveries(), "parity": bool(np.array_equal(got, want))}
        b.close()
    print(row, flush=True)
