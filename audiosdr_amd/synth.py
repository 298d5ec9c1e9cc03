"""Deterministic synthetic I/Q input for tests and benchmarks.

Per channel c, sample index t (continuous over blocks):
    a(t)  = A * (1 + m * sin(2*pi*f_m*t/fs))
    I(t)  = trunc(32767 * (a(t)*cos(phi(t)) + a2*cos(phi2(t)) + n_I(t) + imp_I(t)))
    Q(t)  = trunc(32767 * (a(t)*sin(phi(t)) + a2*sin(phi2(t)) + n_Q(t) + imp_Q(t)))
    phi(t) = 2*pi*f_c*t/fs,  phi2(t) = 2*pi*f_2*t/fs
    n_*    uniform in +-noise from the 32-bit LCG s = s*1664525 + 1013904223 (seed 12345 + c),
           two draws per sample (I then Q), value = ((s >> 8) / 2^24 * 2 - 1) * noise
    imp    (+0.6, -0.5) added to (I, Q) at every t % impulse_every == impulse_every // 2
All arithmetic float64, truncation toward zero, clipped to int16.  Layout [channel][block][128].
"""
import numpy as np

FS = 44100.0
BLOCK = 128
_LCG_A = 1664525
_LCG_C = 1013904223
_M32 = 1 << 32


def _lcg_jump(k):
    """(A_k, C_k) such that s_k = A_k*s_0 + C_k (mod 2^32)."""
    a_k, c_k = 1, 0
    a, c = _LCG_A, _LCG_C
    while k:
        if k & 1:
            a_k, c_k = (a_k * a) % _M32, (c_k * a + c) % _M32
        a, c = (a * a) % _M32, (c * a + c) % _M32
        k >>= 1
    return a_k, c_k


def make_iq(n_channels, n_blocks, fc=6290.0, A=0.25, m=0.0, fm=400.0, noise=0.01, impulse_every=0,
            f2=None, a2=0.0, seed0=12345, start_block=0, channel0=0):
    """Returns (I, Q) int16 arrays of shape [n_channels][n_blocks][128].

    fc / A / f2 may be scalars or per-channel arrays.  `start_block`/`channel0` let a caller
    generate any sub-rectangle of a larger job and get bit-identical samples.
    """
    n = n_blocks * BLOCK
    t0 = start_block * BLOCK
    t = (t0 + np.arange(n, dtype=np.float64))[None, :]
    ch = channel0 + np.arange(n_channels, dtype=np.int64)
    fc = np.broadcast_to(np.asarray(fc, dtype=np.float64), (n_channels,))[:, None]
    A_ = np.broadcast_to(np.asarray(A, dtype=np.float64), (n_channels,))[:, None]
    amp = A_ * (1.0 + m * np.sin(2.0 * np.pi * fm * t / FS))
    ph = 2.0 * np.pi * fc * t / FS
    xi = amp * np.cos(ph)
    xq = amp * np.sin(ph)
    if f2 is not None and a2 != 0.0:
        f2 = np.broadcast_to(np.asarray(f2, dtype=np.float64), (n_channels,))[:, None]
        ph2 = 2.0 * np.pi * f2 * t / FS
        xi = xi + a2 * np.cos(ph2)
        xq = xq + a2 * np.sin(ph2)
    if noise:
        # LCG state after 2*t0 draws, then 2n further draws, vectorised with jump-ahead tables
        a0, c0 = _lcg_jump(2 * t0)
        s0 = ((seed0 + ch) % _M32).astype(np.uint64)
        s0 = (s0 * a0 + c0) % _M32
        ak = np.empty(2 * n, dtype=np.uint64)
        ck = np.empty(2 * n, dtype=np.uint64)
        a_k, c_k = 1, 0
        for k in range(2 * n):
            a_k, c_k = (a_k * _LCG_A) % _M32, (c_k * _LCG_A + _LCG_C) % _M32
            ak[k], ck[k] = a_k, c_k
        s = (s0[:, None] * ak[None, :] + ck[None, :]) % _M32  # uint64 wrap is harmless: 2^32 divides 2^64
        u = ((s >> np.uint64(8)).astype(np.float64) / float(1 << 24)) * 2.0 - 1.0
        xi = xi + noise * u[:, 0::2]
        xq = xq + noise * u[:, 1::2]
    if impulse_every:
        hit = ((t0 + np.arange(n)) % impulse_every) == (impulse_every // 2)
        xi = xi + 0.6 * hit[None, :]
        xq = xq - 0.5 * hit[None, :]
    I = np.clip(np.trunc(32767.0 * xi), -32768, 32767).astype(np.int16)
    Q = np.clip(np.trunc(32767.0 * xq), -32768, 32767).astype(np.int16)
    return I.reshape(n_channels, n_blocks, BLOCK), Q.reshape(n_channels, n_blocks, BLOCK)
