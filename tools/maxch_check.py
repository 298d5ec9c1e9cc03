#!/usr/bin/env python3
"""Maximum batch size of the C ABI (1,048,576 channels) x 3 blocks, USB with one AM and one SAM channel, spot-checked
bit-for-bit against the CPU oracle.  (GPU box; ~2 GB of host memory.)"""
import sys, time, numpy as np
sys.path.insert(0, '.')
import audiosdr_amd as A
from audiosdr_amd.synth import make_iq
from oracle import asdr_oracle as ao
n = 1 << 20
t0 = time.time()
b = A.AudioSDRBatch(n); b.set_launch_timing(True)
b.setDemodMode(A.USBmode); b.enableAudioFilter()
b.setDemodMode(A.AMmode, ch=n - 1); b.setDemodMode(A.SAMmode, ch=12345)
uniq = 4096
I, Q = make_iq(uniq, 3, fc=6290.0, A=0.25, impulse_every=500)
I = np.tile(I, (n // uniq, 1, 1)); Q = np.tile(Q, (n // uniq, 1, 1))
print("setup %.1f s" % (time.time() - t0)); t0 = time.time()
out = b.update(I, Q)
print("update %.1f s, kernel %.3f ms" % (time.time() - t0, b.last_kernel_ms()))
ok = True
for c, mode in ((0, 1), (777, 1), (n - 2, 1), (n - 1, 4), (12345, 5)):
    o = ao.OracleSDR(); o.setDemodMode(mode)
    if mode == 1: o.enableAudioFilter()
    else: o.enableAudioFilter()
    want = o.update(I[c], Q[c]).reshape(3, 128)
    same = np.array_equal(out[c], want); ok = ok and same
    print(c, mode, same)
print("OK" if ok else "MISMATCH")
