#!/usr/bin/env python3
"""Back-to-back asdr_update_device calls on the C2 workload (65,536 channels x 1 block, one stream), in one process:
  period without timing events (one event pair around the whole run / calls),
  period with an event pair around every call, and what those pairs measure (kernel + marker handling).
The difference between a pair's elapsed time and the period is the time between dependent kernels on an in-order HIP stream.
(GPU box.)   python tools/launch_gap.py [calls]
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)


def main():
    import torch
    import audiosdr_amd as A
    from audiosdr_amd.synth import make_iq
    calls = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    n_ch = 65536
    I, Q = make_iq(4096, 2, fc=6290.0, A=0.25)
    I = np.tile(I, (n_ch // 4096, 1, 1)); Q = np.tile(Q, (n_ch // 4096, 1, 1))
    dI = [torch.from_numpy(np.ascontiguousarray(I[:, b])).cuda() for b in range(2)]
    dQ = [torch.from_numpy(np.ascontiguousarray(Q[:, b])).cuda() for b in range(2)]
    dOut = torch.empty((n_ch, 128), dtype=torch.int16, device="cuda")
    b = A.AudioSDRBatch(n_ch)
    b.setDemodMode(A.LSBmode); b.enableAudioFilter()

    def run(k):
        for i in range(k):
            b.update_device(dI[i & 1].data_ptr(), dQ[i & 1].data_ptr(), dOut.data_ptr(), 1, None)
    run(60); torch.cuda.synchronize()
    for rep in range(3):
        b.region_timing_begin(None); run(calls); ms, n = b.region_timing_end()
        p0 = ms / n
        b.kernel_timing_begin(calls); b.region_timing_begin(None); run(calls); ms, n = b.region_timing_end()
        pairs = b.kernel_timing_end(calls)
        p1 = ms / n
        print("period, no events %.4f ms | period with a pair per call %.4f ms | inside the pairs: mean %.4f min %.4f ms" % (p0, p1, float(np.mean(pairs)), float(np.min(pairs))))
    b.close()


if __name__ == "__main__":
    main()
