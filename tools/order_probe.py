import sys, os
sys.path.insert(0, os.getcwd())
mode = sys.argv[1]
import numpy as np
if mode == "lib_first":
    import audiosdr_amd
    b = audiosdr_amd.AudioSDRBatch(8, device=0); b.close()
    import torch
    print("torch after lib:", torch.zeros(4).cuda().sum().item())
elif mode == "torch_first":
    import torch
    print("torch first:", torch.zeros(4).cuda().sum().item())
    import audiosdr_amd
    b = audiosdr_amd.AudioSDRBatch(8, device=0); b.close()
    print("lib after torch ok")
elif mode == "import_torch_then_lib_then_cuda":
    import torch
    import audiosdr_amd
    b = audiosdr_amd.AudioSDRBatch(8, device=0); b.close()
    print("torch after lib (torch imported first):", torch.zeros(4).cuda().sum().item())
