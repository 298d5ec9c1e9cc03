#!/usr/bin/env python3
"""GPU box: N launches of the C2 step on a caller's stream with a given build of the library (a variant from audiosdr_amd/variants/
or the in-tree one) -- the program counter passes profile (tools/pmc_lds_by_phase.sh).
    python3 tools/c2_loop.py [library path | -] [launches] [workload: c2 | c4 | als1]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import audiosdr_amd.binding as _binding

if len(sys.argv) > 1 and sys.argv[1] != "-":
    _binding.library_path = lambda _p=os.path.abspath(sys.argv[1]): _p
import numpy as np
import torch

import audiosdr_amd as A
import bench

N = int(sys.argv[2]) if len(sys.argv) > 2 else 60
wl = sys.argv[3] if len(sys.argv) > 3 else "c2"
n_ch = 65536 if wl == "c2" else 131072
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
if wl == "c2":
    dI, dQ = bench.tiled_input(np, torch, dev, n_ch, 4, n_ch // 4, fc=6290.0, A=0.25)
else:
    dI, dQ = bench.tiled_input(np, torch, dev, n_ch, 4, 4096, fc=6890.0 - 300, A=0.3, m=0.4, f2=7500.0, a2=0.15)
dOut = torch.empty((n_ch, 128), dtype=torch.int16, device=dev)
b = A.AudioSDRBatch(n_ch, device=0)
if wl == "c2":
    bench.configure_c2(b)
elif wl == "c4":
    bench.configure_c4(b)
else:
    b.setDemodMode(1); b.enableALSfilter(); b.setNoiseBlankerThresholdDb(10.0)
for i in range(N):
    b.update_device(dI[i % 4].data_ptr(), dQ[i % 4].data_ptr(), dOut.data_ptr(), 1, stream)
torch.cuda.synchronize()
b.close()
