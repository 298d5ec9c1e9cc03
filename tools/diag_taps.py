#!/usr/bin/env python3
"""Diagnostic (GPU box): mixed-mode batch with >= 512 SAM + ALS channels whose LAST SAM wave is partial; stage taps of its channels vs the oracle.
    ASDR_TOOLS_LIB=audiosdr_amd/variants/libasdr_x.so python3 tools/diag_taps.py"""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _variant  # noqa: E402,F401
import audiosdr_amd as gpu  # noqa: E402
from audiosdr_amd.synth import make_iq  # noqa: E402
from oracle import asdr_oracle as ao  # noqa: E402


def bits(x):
    return np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)


def main():
    n_ch = int(os.environ.get("DIAG_CH", "4235")); n_blk = 6
    I, Q = make_iq(n_ch, n_blk, fc=6890.0 - 300, A=0.3, m=0.4, f2=7500.0, a2=0.15)
    b = gpu.AudioSDRBatch(n_ch)
    b.enable_taps(True)
    for c in range(n_ch):
        b.setDemodMode(c % 7, ch=c)
    b.enableALSfilter(); b.setNoiseBlankerThresholdDb(10.0)
    sam = [c for c in range(n_ch) if c % 7 == 5]
    look = sam[-8:] + sam[:2]
    orcs = {}
    for c in look:
        o = ao.OracleSDR(taps=True); o.setDemodMode(c % 7); o.enableALSfilter(); o.setNoiseBlankerThresholdDb(10.0); orcs[c] = o
    print("SAM channels %d (mod 8 = %d); looking at %s" % (len(sam), len(sam) % 8, look))
    for blk in range(n_blk):
        got = b.update(I[:, blk:blk + 1], Q[:, blk:blk + 1])[:, 0]
        taps = b.read_taps()
        for c in look:
            want = orcs[c].update(I[c, blk], Q[c, blk])
            bad = [t for t in gpu.TAPS if not np.array_equal(bits(taps[t][c]), bits(orcs[c].tap(t)))]
            if bad or not np.array_equal(got[c], want):
                t = bad[0] if bad else None
                extra = ""
                if t:
                    g, w = np.asarray(taps[t][c]).reshape(-1), np.asarray(orcs[c].tap(t)).reshape(-1)
                    i0 = int(np.nonzero(bits(g) != bits(w))[0][0])
                    if t == "ALS":
                        xa = np.asarray(orcs[c].tap("AGC")).reshape(-1)
                        extra = " [x[%d] = %r, x - want = %r]\n    got  %s\n    want %s\n    x    %s\n" % (i0, float(xa[i0]), float(xa[i0] - w[i0]), g[:10].tolist(), w[:10].tolist(), xa[:10].tolist())
                    extra += " first bad tap %s at sample %d: got %r want %r (n bad %d)" % (t, i0, float(g[i0]), float(w[i0]), int((bits(g) != bits(w)).sum()))
                print("block %d ch %d (slot %d of its wave): bad taps %s audio_equal %s%s" % (blk, c, sam.index(c) % 8, bad, np.array_equal(got[c], want), extra), flush=True)
    print("done")


main()
