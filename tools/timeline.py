#!/usr/bin/env python3
"""Phase timeline of single waves of asdr_update_kernel inside a full C2 launch (65,536 channels): a profiling build
(-DASDR_TIMELINE) makes lane 0 of waves 0 / 2731 / 5461 / 8191 write clock64() at 16 phase boundaries into the taps buffer.
Also runs a lone-wave launch (8 channels).  Prints microseconds per phase (s_memtime ticks at 100 MHz).  (GPU box.)

  python tools/timeline.py build   # here
  python tools/timeline.py run     # GPU box
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "audiosdr_amd", "variants", "libasdr_timeline.so")
NAMES = ["prologue+load", "NB: store ring, envelopes", "NB: sequential average/threshold", "NB: mask, ramp, carry, output",
         "IF pipeline", "mixer phase recurrence", "mixer multiply", "Hilbert: stage history", "Hilbert: FIR", "sideband combine",
         "audio pipeline", "AGC: table to LDS", "AGC: sequential envelope", "AGC: apply", "output + status"]


def build():
    from audiosdr_amd import build as b
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    b.build(force=True, extra_flags=["-DASDR_TIMELINE"], out=LIB)
    print("built", LIB)


def run():
    import audiosdr_amd as A
    from audiosdr_amd.synth import make_iq
    import torch
    L = A.binding.load_library(LIB)
    for n_ch in (65536, 8):
        uniq = min(n_ch, 2048)
        I, Q = make_iq(uniq, 6, fc=6290.0, A=0.25)
        I = np.tile(I, (n_ch // uniq, 1, 1)); Q = np.tile(Q, (n_ch // uniq, 1, 1))
        h = L.asdr_create(n_ch, 0)
        L.asdr_setDemodMode(h, -1, 1); L.asdr_enableAudioFilter(h, -1); L.asdr_enable_taps(h, 1)
        dOut = torch.empty((n_ch, 128), dtype=torch.int16, device="cuda")
        rows = []
        dIs = [torch.from_numpy(np.ascontiguousarray(I[:, b])).cuda() for b in range(6)]
        dQs = [torch.from_numpy(np.ascontiguousarray(Q[:, b])).cuda() for b in range(6)]
        warm = int(os.environ.get("TIMELINE_WARM", "300"))   # the AGC envelope has converged, as in bench.py's steady state
        for b in range(warm):
            L.asdr_update_device(h, C.c_void_p(dIs[b % 6].data_ptr()), C.c_void_p(dQs[b % 6].data_ptr()), C.c_void_p(dOut.data_ptr()), 1, None)
        for b in range(6):
            dI, dQ = dIs[(warm + b) % 6], dQs[(warm + b) % 6]
            L.asdr_update_device(h, C.c_void_p(dI.data_ptr()), C.c_void_p(dQ.data_ptr()), C.c_void_p(dOut.data_ptr()), 1, None)
            taps = np.zeros((12, n_ch, 128), dtype=np.float32)
            L.asdr_read_taps(h, taps.ctypes.data_as(C.POINTER(C.c_float)))
            tl = taps.reshape(-1).view(np.uint64)[:128].reshape(4, 32)[:, :16].astype(np.int64)
            if b >= 2:
                rows.append(tl)
        tl = np.stack(rows)                      # [launch][wave][16]
        d = np.diff(tl, axis=2).astype(np.float64)   # ticks per phase
        nw = 4 if n_ch >= 65536 else 1
        d = d[:, :nw]
        tick_us = 0.01                            # s_memtime: 100 MHz on gfx9
        med = np.median(d.reshape(-1, 15), axis=0) * tick_us
        print("== %d channels (%s): wave lifetime %.1f us (median over %d waves x launches)" %
              (n_ch, "full launch" if n_ch > 8 else "lone wave", med.sum(), d.shape[0] * nw))
        for nm, v in zip(NAMES, med):
            print("   %-36s %7.2f us  %5.1f %%" % (nm, v, 100 * v / med.sum()))
        L.asdr_destroy(h)


def stream(T=256, n_ch=512):
    """The block pipeline's three roles of channel group 0 over the last two blocks of a T-block call (WSPR settings, as C5):
    time between two consecutive block starts of each role (its cycle time = the pipeline's time per block if it is the slowest)
    and where inside the block it goes: waits for the neighbours' counters | every phase boundary of the chain."""
    import audiosdr_amd as A
    from audiosdr_amd.synth import make_iq
    import torch
    import bench
    L = A.binding.load_library(LIB)
    I, Q = make_iq(n_ch, T, fc=6890.0, A=0.02, noise=0.05)
    dI, dQ = torch.from_numpy(I).cuda(), torch.from_numpy(Q).cuda()
    dO = torch.empty((n_ch, T, 128), dtype=torch.int16, device="cuda")
    h = L.asdr_create(n_ch, 0)
    class H:   # the bench's C5 settings through the raw handle
        def __getattr__(self, name):
            return lambda *a: getattr(L, "asdr_" + name)(h, -1, *a)
    bench.configure_c5(H())
    L.asdr_enable_taps(h, 1)
    names = ["waits for the neighbours"] + NAMES
    acc = []
    for rep in range(8):
        L.asdr_update_device(h, C.c_void_p(dI.data_ptr()), C.c_void_p(dQ.data_ptr()), C.c_void_p(dO.data_ptr()), T, None)
        L.asdr_synchronize(h)
        taps = np.zeros((12, n_ch, 128), dtype=np.float32)
        L.asdr_read_taps(h, taps.ctypes.data_as(C.POINTER(C.c_float)))
        tl = taps.reshape(-1).view(np.uint64)[:32 * 14].reshape(14, 32).astype(np.int64)
        if rep >= 2:
            acc.append(tl)
    print("pipeline launches:", L.asdr_stream_pipeline_launches(h))
    tl = np.stack(acc)   # [rep][slot][32]
    last, prev = (T - 1) & 1, (T - 2) & 1
    for role in (1, 2, 3):
        a_, p_ = tl[:, 2 * (3 + role) + last], tl[:, 2 * (3 + role) + prev]
        cyc = np.median(a_[:, 16] - p_[:, 16])
        print("== role %d: %.0f cycles between two block starts" % (role, cyc))
        seq = np.concatenate([a_[:, 16:17], a_[:, :16]], axis=1)    # pre-wait, TL0..TL15
        d = np.diff(seq, axis=1)
        med = np.median(d, axis=0)
        for nm, v in zip(names, med):
            if v > 0 and v < 10 * cyc:
                print("   %-36s %8.0f cycles" % (nm, v))
    L.asdr_destroy(h)


if __name__ == "__main__":
    if sys.argv[1:2] == ["build"]:
        build()
    elif sys.argv[1:2] == ["stream"]:
        stream(*[int(x) for x in sys.argv[2:]])
    else:
        run()
