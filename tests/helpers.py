"""Shared helpers for the parity tests (setter scripts applied identically to the HIP batch and to the oracle)."""
import numpy as np


def S(method, *args, sel=None):
    return (method, args, sel)


def apply_setters(batch, oracles, setters):
    n = len(oracles)
    for meth, args, sel in setters:
        for c in range(n):
            if sel is None or sel(c):
                if batch is not None:
                    getattr(batch, meth)(*args, ch=c)
                getattr(oracles[c], meth)(*args)


def f32_bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def compare_status(A, batch, oracles):
    st = batch.read_status()
    for c, o in enumerate(oracles):
        assert int(st["agc_active"][c]) == o.AGCisActive(), "AGCisActive ch %d" % c
        assert int(st["nb_detected"][c]) == o.NoiseBlankerDetection(), "NoiseBlankerDetection ch %d" % c
        assert int(st["sam_locked"][c]) == o.getSAMphaseLockStatus(), "SAM lock ch %d" % c
        assert f32_bits(st["sam_frequency"][c]) == f32_bits(np.float32(o.getSAMfrequency())), "SAM frequency ch %d" % c
        assert f32_bits(st["am_carrier"][c]) == f32_bits(np.float32(o.getAMcarrierLevel())), "carrier ch %d" % c
