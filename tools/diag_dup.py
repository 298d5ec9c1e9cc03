#!/usr/bin/env python3
"""Diagnostic (GPU box): the C4 share of tests/test_gpu_full_size.py; which channels differ from their duplicates, in which block, by how much.
    ASDR_TOOLS_LIB=audiosdr_amd/variants/libasdr_x.so python3 tools/diag_dup.py [n_blk] [repeat]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _variant  # noqa: E402,F401
import torch  # noqa: E402
import audiosdr_amd as gpu  # noqa: E402
from audiosdr_amd.synth import make_iq  # noqa: E402


def main():
    n_blk = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    rep = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    sync = os.environ.get("DIAG_SYNC", "1") == "1"
    n_ch, uniq = 131072, 3584
    I, Q = make_iq(uniq, n_blk, fc=6890.0 - 300, A=0.3, m=0.4, f2=7500.0, a2=0.15)
    reps = (n_ch + uniq - 1) // uniq
    dI = [torch.from_numpy(np.ascontiguousarray(I[:, b])).cuda().repeat(reps, 1)[:n_ch].contiguous() for b in range(n_blk)]
    dQ = [torch.from_numpy(np.ascontiguousarray(Q[:, b])).cuda().repeat(reps, 1)[:n_ch].contiguous() for b in range(n_blk)]
    torch.cuda.synchronize()
    for r in range(rep):
        batch = gpu.AudioSDRBatch(n_ch)
        L = gpu.load_library()
        for c in range(n_ch):
            L.asdr_setDemodMode(batch._h, c, c % 7)
        batch.enableALSfilter(); batch.setNoiseBlankerThresholdDb(10.0)
        dOut = torch.empty((n_ch, 128), dtype=torch.int16, device="cuda")
        for b in range(n_blk):
            batch.update_device(dI[b].data_ptr(), dQ[b].data_ptr(), dOut.data_ptr(), 1, 0)
            if sync:
                batch.synchronize()
            else:
                torch.cuda.synchronize()
            out = dOut.cpu().numpy()
            ref = np.tile(out[:uniq], (reps, 1))[:n_ch]
            bad = np.nonzero((out != ref).any(axis=1))[0]
            if len(bad):
                d = np.abs(out[bad].astype(np.int32) - ref[bad].astype(np.int32))
                print("run %d block %d: %d channels differ; first %s; modes %s; max |diff| %d; first differing sample idx %s" % (
                    r, b, len(bad), bad[:12].tolist(), sorted(set((bad % 7).tolist())), int(d.max()),
                    [int(np.nonzero(out[c] != ref[c])[0][0]) for c in bad[:6]]), flush=True)
                print("   waves (channel // 8) of the differing channels: %s" % sorted(set((bad // 8).tolist()))[:20], flush=True)
        batch.close()
    print("done")


main()
