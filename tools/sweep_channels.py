#!/usr/bin/env python3
"""Kernel time vs batch size (waves = channels/8) to expose the resident-wave slots per CU (GPU box)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import audiosdr_amd as A
from audiosdr_amd.synth import make_iq
import torch
I, Q = make_iq(2048, 1, fc=6290.0, A=0.25)
for waves_per_cu in [1, 2, 4, 6, 8, 10, 11, 12, 13, 14, 16, 18, 20, 22, 24, 25, 26, 28, 30, 32, 34, 36, 40, 48]:
    n_ch = 256 * waves_per_cu * 8
    reps = (n_ch + 2047) // 2048
    dI = torch.from_numpy(np.ascontiguousarray(np.tile(I, (reps, 1, 1))[:n_ch, 0])).cuda()
    dQ = torch.from_numpy(np.ascontiguousarray(np.tile(Q, (reps, 1, 1))[:n_ch, 0])).cuda()
    dO = torch.empty((n_ch, 128), dtype=torch.int16, device="cuda")
    b = A.AudioSDRBatch(n_ch); b.set_launch_timing(True)
    b.setDemodMode(1); b.enableAudioFilter()
    ts = []
    for i in range(12):
        b.update_device(dI.data_ptr(), dQ.data_ptr(), dO.data_ptr(), 1, 0)
        ts.append(b.last_kernel_ms())
    t = float(np.median(ts[4:]))
    print("waves/CU %2d  channels %6d  kernel %.4f ms  -> %.1f Mblocks/s" % (waves_per_cu, n_ch, t, n_ch / t / 1e3))
    b.close()
