#!/usr/bin/env python3
"""Development aid: run many configurations through the HIP path and the CPU oracle and report, per
configuration, the first stage tap and block at which they diverge.  (GPU box only.)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import audiosdr_amd as A  # noqa: E402
from audiosdr_amd.synth import make_iq  # noqa: E402
from oracle import asdr_oracle as ao  # noqa: E402


def run_case(name, n_ch, n_blk, setters, sig, per_block=True):
    """setters: list of (method, args) applied to both; sig: kwargs for make_iq."""
    I, Q = make_iq(n_ch, n_blk, **sig)
    batch = A.AudioSDRBatch(n_ch)
    batch.enable_taps(True)
    orcs = [ao.OracleSDR(taps=True) for _ in range(n_ch)]
    for meth, args, chsel in setters:
        for c in range(n_ch):
            if chsel is None or chsel(c):
                getattr(batch, meth)(*args, ch=c)
                getattr(orcs[c], meth)(*args)
    first_bad = None
    n_bad_total = 0
    for b in range(n_blk):
        got = batch.update(I[:, b:b + 1], Q[:, b:b + 1])[:, 0]
        taps = batch.read_taps()
        for c in range(n_ch):
            want = orcs[c].update(I[c, b], Q[c, b])
            nbad = int((got[c] != want).sum())
            n_bad_total += nbad
            if nbad and first_bad is None:
                where = "out"
                for t in A.TAPS:
                    g, w = taps[t][c], orcs[c].tap(t)
                    if not np.array_equal(g.view(np.uint32), w.view(np.uint32)):
                        i = int(np.nonzero(g.view(np.uint32) != w.view(np.uint32))[0][0])
                        where = "%s[%d] gpu=%r cpu=%r" % (t, i, float(g[i]), float(w[i]))
                        break
                first_bad = "block %d ch %d: %d bad samples; first diverging tap: %s" % (b, c, nbad, where)
    st = batch.read_status()
    for c in range(n_ch):
        o = orcs[c]
        exp = (o.AGCisActive(), o.NoiseBlankerDetection(), o.getSAMphaseLockStatus())
        gotst = (int(st["agc_active"][c]), int(st["nb_detected"][c]), int(st["sam_locked"][c]))
        if exp != gotst and first_bad is None:
            first_bad = "status mismatch ch %d: gpu %r cpu %r" % (c, gotst, exp)
        if (np.float32(o.getSAMfrequency()) != st["sam_frequency"][c] or np.float32(o.getAMcarrierLevel()) != st["am_carrier"][c]) \
                and first_bad is None:
            first_bad = "float status mismatch ch %d: gpu (%r,%r) cpu (%r,%r)" % (
                c, st["sam_frequency"][c], st["am_carrier"][c], o.getSAMfrequency(), o.getAMcarrierLevel())
    batch.close()
    print("%-28s %s" % (name, "OK (bit-exact, %d ch x %d blk)" % (n_ch, n_blk) if first_bad is None and n_bad_total == 0
                         else "FAIL total_bad=%d  %s" % (n_bad_total, first_bad)))
    return first_bad is None and n_bad_total == 0


def main():
    ok = True
    tone = dict(fc=6290.0, A=0.25)
    am = dict(fc=6890.0, A=0.3, m=0.5, fm=400.0)
    imp = dict(fc=6290.0, A=0.25, impulse_every=900)
    S = lambda m, *a, sel=None: (m, a, sel)
    cases = [
        ("usb_nb_off_agc_off", 3, 6, [S("setDemodMode", A.USBmode), S("disableNoiseBlanker"), S("disableAGC")], tone),
        ("usb_nb_off", 3, 6, [S("setDemodMode", A.USBmode), S("disableNoiseBlanker")], tone),
        ("usb_default_nb", 3, 8, [S("setDemodMode", A.USBmode)], tone),
        ("usb_c2", 9, 10, [S("setDemodMode", A.USBmode), S("enableAudioFilter")], imp),
        ("lsb_c2", 2, 8, [S("setDemodMode", A.LSBmode), S("enableAudioFilter")], imp),
        ("cw_usb", 2, 8, [S("setDemodMode", A.CW_USBmode), S("enableAudioFilter"), S("setAudioFilter", A.audioCW)], dict(fc=6390 + 700.0, A=0.2)),
        ("cw_lsb", 2, 8, [S("setDemodMode", A.CW_LSBmode), S("setNoiseBlankerThresholdDb", 10.0)], dict(fc=7390 - 700.0, A=0.2)),
        ("wspr_sketch", 2, 8, [S("enableAGC"), S("setAGCmode", A.AGCmedium), S("disableALSfilter"), S("disableNoiseBlanker"),
                               S("setNoiseBlankerThresholdDb", 10.0), S("setInputGain", 1.0), S("setOutputGain", 0.5),
                               S("setIQgainBalance", 1.020), S("setAudioFilter", A.audioWSPR), S("setDemodMode", A.WSPRmode),
                               S("setMute", 0)], dict(fc=6890.0, A=0.02, noise=0.05)),
        ("am_default", 2, 8, [S("setDemodMode", A.AMmode)], am),
        ("am_nb10", 2, 10, [S("setDemodMode", A.AMmode), S("setNoiseBlankerThresholdDb", 10.0), S("enableAudioFilter"), S("setAudioFilter", A.audioAM)], am),
        ("sam_nb10", 4, 12, [S("setDemodMode", A.SAMmode), S("setNoiseBlankerThresholdDb", 10.0), S("enableAudioFilter"), S("setAudioFilter", A.audioAM)], am),
        ("sam_default_unlocked", 2, 8, [S("setDemodMode", A.SAMmode)], am),
        ("usb_als_notch", 2, 8, [S("setDemodMode", A.USBmode), S("setNoiseBlankerThresholdDb", 10.0), S("enableALSfilter")],
         dict(fc=6290.0, A=0.25, f2=7290.0, a2=0.125)),
        ("usb_als_peak_static", 2, 6, [S("setDemodMode", A.USBmode), S("disableNoiseBlanker"), S("enableALSfilter"), S("setALSfilterPeak"),
                                       S("setALSfilterStatic")], tone),
        ("usb_als_params", 2, 8, [S("setDemodMode", A.USBmode), S("disableNoiseBlanker"), S("enableALSfilter"),
                                  S("setALSfilterParams", 100, 0.25, 7.0)], dict(fc=6290.0, A=0.25, f2=7290.0, a2=0.125)),
        ("mixed_modes_als", 21, 10, [S("setNoiseBlankerThresholdDb", 10.0), S("enableALSfilter")] +
         [S("setDemodMode", m, sel=(lambda c, m=m: c % 7 == m)) for m in range(7)], dict(fc=6890.0 - 300, A=0.3, m=0.4, f2=7500.0, a2=0.1)),
        ("muted_gain", 2, 4, [S("setDemodMode", A.USBmode), S("setMute", 1)], tone),
        ("gains", 2, 6, [S("setDemodMode", A.USBmode), S("setInputGain", 3.3), S("setOutputGain", 0.9), S("setAGCstaticGain", 25.0)], tone),
        ("agc_fast_thresh", 2, 8, [S("setDemodMode", A.USBmode), S("setAGCmode", A.AGCfast), S("setAGCthreshold", -40.0), S("setAGCslope", 0.3),
                                   S("setAGCkneeWidth", 6.0)], tone),
    ]
    only = sys.argv[1:]
    t0 = time.time()
    for c in cases:
        if only and c[0] not in only:
            continue
        ok &= run_case(*c)
    print("ALL OK" if ok else "SOME FAILED", "(%.1fs)" % (time.time() - t0))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
