import os, sys, time, json
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import audiosdr_amd as A
from audiosdr_amd.synth import make_iq
T = 128
for n_ch in (512,):
    I, Q = make_iq(n_ch, T, fc=6890.0 - 300, A=0.3, m=0.4, noise=0.01)
    dI, dQ = torch.from_numpy(I).cuda(), torch.from_numpy(Q).cuda()
    dO = torch.empty((n_ch, T, 128), dtype=torch.int16, device="cuda")
    for use_stream in (0, 1):
        st = torch.cuda.Stream() if use_stream else None
        h = st.cuda_stream if st else 0
        b = A.AudioSDRBatch(n_ch)
        if os.environ.get("BENCH_SAM"):
            b.setDemodMode(5); b.setNoiseBlankerThresholdDb(10.0); b.enableAudioFilter(); b.setAudioFilter(0)
        else:
            b.setDemodMode(1); b.enableALSfilter(); b.setNoiseBlankerThresholdDb(10.0)
        torch.cuda.synchronize()
        for _ in range(2): b.update_device(dI.data_ptr(), dQ.data_ptr(), dO.data_ptr(), T, h)
        b.synchronize()
        t0 = time.perf_counter()
        for _ in range(4): b.update_device(dI.data_ptr(), dQ.data_ptr(), dO.data_ptr(), T, h)
        b.synchronize()
        ms = (time.perf_counter() - t0) / 4 * 1e3
        print(json.dumps({"stream": "created" if use_stream else "null", "us_per_block": round(ms * 1e3 / T, 2), "als_role_calls": b.als_role_calls(), "sam_chunk_calls": b.sam_chunk_calls()}), flush=True)
        b.close()
