#!/usr/bin/env python3
"""Replays tests/test_gpu_fuzz.py::test_fuzz_control_surface[seed] with stage taps; prints the first diverging tap (GPU box)."""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import audiosdr_amd as A
from audiosdr_amd.synth import make_iq
from oracle import asdr_oracle as ao
from helpers import S, apply_setters
from test_gpu_fuzz import _random_setter

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rng = np.random.default_rng(seed)
n_ch, n_blk = 40, 36
fc = 6890.0 + rng.uniform(-1800, 1800, n_ch)
I, Q = make_iq(n_ch, n_blk, fc=fc, A=rng.uniform(0.01, 0.6, n_ch), m=0.4, fm=300.0, impulse_every=int(rng.integers(300, 900)), f2=fc + 700.0, a2=0.05)
batch = A.AudioSDRBatch(n_ch); batch.enable_taps(True)
orcs = [ao.OracleSDR(taps=True) for _ in range(n_ch)]
log = {c: [] for c in range(n_ch)}
def do(meth, args, mask):
    apply_setters(batch, orcs, [S(meth, *args, sel=lambda c, m=mask: bool(m[c]))])
    for c in range(n_ch):
        if mask[c]: log[c].append((meth,) + tuple(args))
for _ in range(60):
    meth, args, _s = _random_setter(rng); mask = rng.random(n_ch) < 0.3; do(meth, args, mask)
for b in range(n_blk):
    for _ in range(int(rng.integers(0, 4))):
        meth, args, _s = _random_setter(rng); mask = rng.random(n_ch) < 0.2; do(meth, args, mask)
        for c in range(n_ch):
            if mask[c]: log[c].append(("--- before block %d" % b,))
    got = batch.update(I[:, b:b + 1], Q[:, b:b + 1])[:, 0]
    taps = batch.read_taps()
    for c in range(n_ch):
        want = orcs[c].update(I[c, b], Q[c, b])
        if not np.array_equal(got[c], want):
            print("MISMATCH block", b, "ch", c, "mode", orcs[c].getDemodMode(), "nbad", int((got[c] != want).sum()))
            for t in A.TAPS:
                g, w = taps[t][c], orcs[c].tap(t)
                if not np.array_equal(g.view(np.uint32), w.view(np.uint32)):
                    i = int(np.nonzero(g.view(np.uint32) != w.view(np.uint32))[0][0])
                    print("  first diverging tap", t, "index", i, "gpu", float(g[i]), "cpu", float(w[i]), "n", int((g.view(np.uint32) != w.view(np.uint32)).sum()))
                    break
            o = orcs[c]
            print("  flags: nb", o.NoiseBlankerisEnabled(), "af", o.getAudioFilter(), "agc", o.AGCisEnabled(), "als", o.ALSfilterIsEnabled(), "notch", o.ALSfilterIsNotch(), "adaptive", o.ALSfilterIsAdaptive())
            print("  setter log:", log[c])
            sys.exit(1)
print("no mismatch")
