#!/usr/bin/env python3
"""bench.py -- Msamples/s through the full SSB demod chain, batched 128-sample blocks (BASELINE.json).

Workload at every N (weak scaling; channels shard embarrassingly, no collective):
  SURVEY.md 8d config C2 per GPU: 65,536 independent channels x one 128-sample block per step,
  USB demodulation, noise blanker on (default threshold 1.2), IF band-pass, complex mixer,
  257-tap Hilbert, audio IIR filter enabled (bw2700), AGC default, ALS off.
  int16 I/Q rows and the int16 audio rows are resident in HBM before the timed region.
A "step" is one asdr_update_device() call = one pass of the hot path over the whole batch.

Prints ONE JSON line (rank 0).  `value` = samples processed by all ranks / max-over-ranks wall time.
`roofline` is for the single kernel of the path (asdr_update_kernel): algorithmic bytes per launch
(10,520 B per channel-block: SURVEY.md 8d) / its mean duration from HIP events recorded on the launch
stream around every launch of the timed region.  `cpu_baseline` is the CPU oracle (a port, the reference
is unbuildable here) timed on this host on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CHANNELS_PER_GPU = 65536
BLOCK = 128
ALGO_BYTES_PER_BLOCK = 10520          # 768 I/O + 96 params + 2 x 4828 carried state (C2)
ALGO_READ_BYTES_PER_BLOCK = 5436      # HBM-read share of the above
HBM_PEAK_GBS = 8000.0                 # MI355X_MICROARCH.md: 8.0 TB/s spec
N_INPUT_BLOCKS = 4                    # distinct resident input blocks cycled through by the steps


def configure_c2(sdr):
    sdr.setDemodMode(1)        # USBmode
    sdr.enableAudioFilter()    # bw2700 from init(); NB + AGC are on by default


def cpu_baseline(seconds_budget=12.0):
    """Oracle (port) on the host cores: bounded sample of the C2 workload."""
    from audiosdr_amd.synth import make_iq
    from oracle import asdr_oracle as ao
    cores = os.cpu_count() or 1
    I, Q = make_iq(64, 64, fc=6290.0, A=0.25)
    t1, _ = ao.bench_run(0, I, Q, 1)                       # calibrate: 1 core
    rate1 = I.size / t1
    n_ch = int(max(cores, min(8192, rate1 * cores * (seconds_budget * 0.5) / (64 * BLOCK))))
    n_ch -= n_ch % cores
    I, Q = make_iq(n_ch, 64, fc=6290.0, A=0.25)
    tn, _ = ao.bench_run(0, I, Q, cores)
    return {"value": round(I.size / tn / 1e6, 3), "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": "CPU oracle (oracle/asdr_oracle.c, gcc -O2 -ffp-contract=off), C2 USB chain, %d channels x 64 blocks "
                      "on %d threads; 1-thread rate %.2f Msamples/s" % (n_ch, cores, rate1 / 1e6)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--channels", type=int, default=CHANNELS_PER_GPU, help="channels per GPU (default = C2)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1 or "RANK" in os.environ:   # launched by torch.distributed.run: one process per GPU, RCCL for barrier/max only
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import audiosdr_amd as A
    from audiosdr_amd.synth import make_iq

    n_ch = args.channels
    # this rank's shard of the job: channels [rank*n_ch, (rank+1)*n_ch); a quarter is generated and tiled
    uniq = max(8, n_ch // 4)
    I, Q = make_iq(uniq, N_INPUT_BLOCKS, fc=6290.0, A=0.25, channel0=rank * n_ch)
    reps = (n_ch + uniq - 1) // uniq
    I = np.tile(I, (reps, 1, 1))[:n_ch]
    Q = np.tile(Q, (reps, 1, 1))[:n_ch]
    # resident layout per step: [channel][1 block][128]
    dI = [torch.from_numpy(np.ascontiguousarray(I[:, b])).to(dev) for b in range(N_INPUT_BLOCKS)]
    dQ = [torch.from_numpy(np.ascontiguousarray(Q[:, b])).to(dev) for b in range(N_INPUT_BLOCKS)]
    dOut = torch.empty((n_ch, BLOCK), dtype=torch.int16, device=dev)

    batch = A.AudioSDRBatch(n_ch, device=local_rank)
    configure_c2(batch)
    stream = torch.cuda.current_stream().cuda_stream

    def step(i):
        b = i % N_INPUT_BLOCKS
        batch.update_device(dI[b].data_ptr(), dQ[b].data_ptr(), dOut.data_ptr(), 1, stream)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    fence()
    # one HIP-event pair per launch, recorded by the library on the launch stream inside the timed region
    batch.kernel_timing_begin(args.steps)
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    fence()
    t1 = time.perf_counter()
    wall = t1 - t0
    kernel_ms = batch.kernel_timing_end(args.steps)
    if dist is not None:
        tw = torch.tensor([wall], dtype=torch.float64, device=dev)
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        wall = float(tw.item())

    if rank == 0:
        samples = float(world) * n_ch * BLOCK * args.steps
        k_ms = float(np.mean(kernel_ms))
        ach = ALGO_BYTES_PER_BLOCK * n_ch / (k_ms * 1e-3) / 1e9
        traffic = None   # HBM bytes per launch from the committed rocprofv3 PMC passes of this same command (tools/prof_pmc.sh)
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_latest.json")) as f:
                pj = json.load(f)
            if n_ch == CHANNELS_PER_GPU:
                traffic = pj.get("traffic_bytes_per_launch")
        except Exception:
            pass
        out = {
            "metric": "Msamples/s through full SSB demod chain, batched 128-sample blocks",
            "value": round(samples / wall / 1e6, 2), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(wall / args.steps * 1e3, 5), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "C2: SSB (USB) demod, %d channels/GPU x 1 block/step, NB+IF+mixer+Hilbert+audio IIR+AGC" % n_ch,
                       "channels_per_gpu": n_ch, "block": BLOCK, "sharding": "channels, no collective",
                       "input": "tone + LCG noise per channel, no impulses: the blanker runs (envelope, average, threshold) but detects "
                                "nothing, so its mask stays all ones; with an impulse in every block of every channel the same chain "
                                "is ~8 % slower (tools/bench_configs.py c2)"},
            "roofline": {"bound": "hbm", "kernel": "asdr_update_kernel", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "kernel_ms": round(k_ms, 5), "algorithmic_bytes_per_launch": ALGO_BYTES_PER_BLOCK * n_ch,
                         "hbm_read_share_frac": round(ALGO_READ_BYTES_PER_BLOCK * n_ch / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    batch.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
