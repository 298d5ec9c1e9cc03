#!/usr/bin/env python3
"""Kernel time of the blocks around the hot path at the C2 batch size (65,536 channels), HIP-event timed, each
spot-checked against the CPU oracle.  One JSON line per case with the HBM roofline fraction (algorithmic bytes:
int16 in + int16 out per channel-block; state traffic excluded).  (GPU box.)

  python tools/bench_front.py [n_channels] [blocks_per_launch]
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _variant  # noqa: E402,F401  (ASDR_TOOLS_LIB)
import audiosdr_amd as A  # noqa: E402
from oracle import asdr_oracle as ao  # noqa: E402
from test_front_oracle import tone_iq  # noqa: E402
import torch  # noqa: E402

HBM_PEAK = 8.0e12


def med(batch, fn, n=12, warm=3):
    ms = []
    for i in range(n + warm):
        fn()
        t = batch.last_kernel_ms()
        if i >= warm:
            ms.append(t)
    return float(np.median(ms))


def line(name, n_ch, T, ms, bytes_per_block, parity, extra=None):
    gbs = n_ch * T * bytes_per_block / (ms * 1e-3) / 1e9
    out = {"case": name, "channels": n_ch, "blocks_per_launch": T, "kernel_ms_median": round(ms, 5),
           "Msamples_per_s": round(n_ch * T * 128 / ms / 1e3, 1),
           "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": round(gbs * 1e9 / HBM_PEAK, 4)},
           "algorithmic_bytes_per_block": bytes_per_block, "parity": bool(parity)}
    out.update(extra or {})
    print(json.dumps(out), flush=True)


def main():
    n_ch = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    uniq = 64
    streams = [tone_iq(T + 2, 6000.0 + 97.0 * c, amp=0.25, q_delay=c % 3 - 1, noise=0.003, seed=c) for c in range(uniq)]
    I = np.stack([s[0] for s in streams]).reshape(uniq, T + 2, 128)
    Q = np.stack([s[1] for s in streams]).reshape(uniq, T + 2, 128)
    reps = n_ch // uniq
    Ib, Qb = np.tile(I, (reps, 1, 1)), np.tile(Q, (reps, 1, 1))
    dI, dQ = torch.from_numpy(Ib).cuda(), torch.from_numpy(Qb).cuda()
    dOi, dOq = torch.empty_like(dI), torch.empty_like(dQ)
    S = T + 2

    # ---- pre-processor: fixed correction (+1) and swap; then detector running
    for detect in (False, True):
        p = A.AudioSDRpreProcessorBatch(n_ch)
        orc = [ao.OraclePreProcessor() for _ in range(4)]
        if detect:
            p.startAutoI2SerrorDetection()
            [o.startAutoI2SerrorDetection() for o in orc]
        else:
            p.setI2SerrorCompensation(1); p.swapIQ(True)
            [(o.setI2SerrorCompensation(1), o.swapIQ(True)) for o in orc]
        p.update_device(dI.data_ptr(), dQ.data_ptr(), dOi.data_ptr(), dOq.data_ptr(), T, S, S)
        p.synchronize()
        gi, gq = dOi[:4, :T].cpu().numpy(), dOq[:4, :T].cpu().numpy()
        ok = True
        for c in range(4):
            wi, wq = orc[c].update(I[c, :T], Q[c, :T])
            ok = ok and np.array_equal(gi[c].reshape(-1), wi) and np.array_equal(gq[c].reshape(-1), wq)
        ms = med(p, lambda: p.update_device(dI.data_ptr(), dQ.data_ptr(), dOi.data_ptr(), dOq.data_ptr(), T, S, S))
        line("AudioSDRpreProcessor, " + ("detector on (128-pt FFT per block)" if detect else "correction +1 and swap"), n_ch, T, ms, 1024, ok)
        p.close()

    # ---- IQ generator (real in, I/Q out)
    g = A.AudioIQgeneratorBatch(n_ch)
    og = [ao.OracleIQgenerator() for _ in range(4)]
    g.update_device(dI.data_ptr(), dOi.data_ptr(), dOq.data_ptr(), T, S, S)
    g.synchronize()
    gi, gq = dOi[:4, :T].cpu().numpy(), dOq[:4, :T].cpu().numpy()
    ok = True
    for c in range(4):
        wi, wq = og[c].update(I[c, :T])
        ok = ok and np.array_equal(gi[c].reshape(-1), wi) and np.array_equal(gq[c].reshape(-1), wq)
    ms = med(g, lambda: g.update_device(dI.data_ptr(), dOi.data_ptr(), dOq.data_ptr(), T, S, S))
    # The generator is bound by its FIR, not by HBM: 128 outputs x 64 folded taps x (subtract, multiply, add) = 24,576 separately
    # rounded FP32 operations per channel-block (no FMA: parity with AudioIQgenerator.cpp:60-76), against the vector unit's
    # 157.3 TFLOP/s / 2 = 78.6 T operations/s without FMA (MI355X_MICROARCH.md: 64 FLOP/clk/SIMD counts an FMA as two).
    tops = n_ch * T * 24576 / (ms * 1e-3) / 1e12
    line("AudioIQgenerator (257-tap Hilbert)", n_ch, T, ms, 768, ok,
         {"state_bytes_per_launch_per_channel": 768 if T == 1 else 1024,   # raw int16 ring: two blocks read, one (a multi-block call: two) written
          "roofline_valu": {"bound": "valu", "achieved": round(tops, 2), "peak": 78.6, "unit": "T FP32 operations/s (no FMA)", "frac": round(tops / 78.6, 4),
                            "operations_per_channel_block": 24576}})
    g.close()

    # ---- grabber (two blocks -> 256 interleaved complex samples)
    gr = A.AudioGrabberComplex256Batch(n_ch)
    gr.update_device(dI.data_ptr(), dQ.data_ptr(), 2, S)
    ogr = ao.OracleGrabber(); ogr.update(I[1, :2], Q[1, :2])
    ok = np.array_equal(gr.grab(1)[1], ogr.grab())
    ev = []
    L = A.load_library()
    for i in range(10):
        s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s0.record()
        gr.update_device(dI.data_ptr(), dQ.data_ptr(), 2, S, stream=torch.cuda.current_stream().cuda_stream)
        s1.record(); torch.cuda.synchronize()
        ev.append(s0.elapsed_time(s1))
    line("AudioGrabberComplex256 (2 blocks -> 1 buffer)", n_ch, 2, float(np.median(ev[3:])), 1024 + 1024, ok)
    gr.close()


if __name__ == "__main__":
    main()
