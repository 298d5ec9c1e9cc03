#!/usr/bin/env python3
"""Regenerate tests/golden/chain_vectors.npz.

These are REGRESSION fixtures produced by this repository's own CPU oracle (oracle/asdr_oracle.c) -- NOT by the
reference, which ships no vectors and cannot be built here.  They pin today's oracle behaviour so that
(a) an accidental change of the oracle is caught on CPU and (b) the HIP path can be checked on the GPU box against
committed expected outputs as well as against the live oracle.  Inputs are regenerated from audiosdr_amd.synth
(deterministic), only outputs are stored.
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(HERE, ".."))
from cases import CASES  # noqa: E402
from helpers import apply_setters  # noqa: E402
from audiosdr_amd.synth import make_iq  # noqa: E402
from oracle import asdr_oracle as ao  # noqa: E402


def run_case(name):
    n_ch, n_blk, setters, sig = CASES[name]
    I, Q = make_iq(n_ch, n_blk, **sig)
    orcs = [ao.OracleSDR() for _ in range(n_ch)]
    apply_setters(None, orcs, setters)
    out = np.stack([orcs[c].update(I[c], Q[c]).reshape(n_blk, 128) for c in range(n_ch)])
    status = np.array([[o.AGCisActive(), o.NoiseBlankerDetection(), o.getSAMphaseLockStatus()] for o in orcs], dtype=np.int32)
    fstat = np.array([[o.getSAMfrequency(), o.getAMcarrierLevel()] for o in orcs], dtype=np.float32)
    return out, status, fstat


def main():
    data = {}
    for name in sorted(CASES):
        out, status, fstat = run_case(name)
        data[name + "/out"] = out
        data[name + "/status"] = status
        data[name + "/fstatus"] = fstat
        print("%-32s %s sha1=%s" % (name, out.shape, hashlib.sha1(out.tobytes()).hexdigest()[:12]))
    np.savez_compressed(os.path.join(HERE, "chain_vectors.npz"), **data)


if __name__ == "__main__":
    main()
