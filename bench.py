#!/usr/bin/env python3
"""bench.py -- Msamples/s through the full SSB demod chain, batched 128-sample blocks (BASELINE.json).

Workload at every N (weak scaling; channels shard embarrassingly, no collective):
  SURVEY.md 8d config C2 per GPU: 65,536 independent channels x one 128-sample block per step,
  USB demodulation, noise blanker on (default threshold 1.2), IF band-pass, complex mixer,
  257-tap Hilbert, audio IIR filter enabled (bw2700), AGC default, ALS off.
  int16 I/Q rows and the int16 audio rows are resident in HBM before the timed region.
A "step" is one asdr_update_device() call = one pass of the hot path over the whole batch.

Launching:
  python bench.py --gpus N ...            N > 1 without a distributed launcher: this process starts
                                          `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD
                                          (before anything here touches torch or the GPU) and relays rank 0's JSON line
  python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...   (what the driver does for N > 1)
  A launcher whose WORLD_SIZE differs from --gpus is an error (exit 2), never a silently mislabelled line.
  --dry-run: no GPU and no HIP library: gloo backend, the timed loop is empty; checks the launch / reduce / JSON plumbing.

Prints ONE JSON line (rank 0).  `value` = samples processed by all ranks / max-over-ranks wall time.
`roofline` is for the single kernel of the path (asdr_update_kernel): algorithmic bytes per launch
(10,520 B per channel-block: SURVEY.md 8d) / its mean duration from HIP events recorded on the launch
stream around every launch of the timed region.  `cpu_baseline` is the CPU oracle (a port, the reference
is unbuildable here) timed on this host on a bounded sample of the same workload.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CHANNELS_PER_GPU = 65536
BLOCK = 128
ALGO_BYTES_PER_BLOCK = 10520          # 768 I/O + 96 params + 2 x 4828 carried state (C2)
ALGO_READ_BYTES_PER_BLOCK = 5436      # HBM-read share of the above
HBM_PEAK_GBS = 8000.0                 # MI355X_MICROARCH.md: 8.0 TB/s spec
N_INPUT_BLOCKS = 4                    # distinct resident input blocks cycled through by the steps
PMC_FILE = os.path.join("profiles", "pmc_latest.json")


def configure_c2(sdr):
    sdr.setDemodMode(1)        # USBmode
    sdr.enableAudioFilter()    # bw2700 from init(); NB + AGC are on by default


SETTLE_LAUNCHES = 1500    # untimed launches in front of the timed region (>= --warmup): clocks settled, see main()


def cpu_info():
    """CPU model and physical core count of this host (SURVEY.md 8d asks for both next to the baseline)."""
    model, phys = None, set()
    try:
        pid = cid = None
        with open("/proc/cpuinfo") as f:
            for line in f:
                k, _, v = line.partition(":")
                k, v = k.strip(), v.strip()
                if k == "model name" and model is None:
                    model = v
                elif k == "physical id":
                    pid = v
                elif k == "core id":
                    cid = v
                elif not k and pid is not None and cid is not None:
                    phys.add((pid, cid)); pid = cid = None
        if pid is not None and cid is not None:
            phys.add((pid, cid))
    except OSError:
        pass
    return model, (len(phys) or None)


def cpu_baseline(seconds_budget=14.0):
    """Oracle (port) on the host cores: bounded sample of the C2 workload.  Instances are created and configured before
    the timed region (oracle/asdr_oracle.c ao_bench_run); the all-thread sample is sized for >= 2 s of wall time."""
    from audiosdr_amd.synth import make_iq
    from oracle import asdr_oracle as ao
    threads = os.cpu_count() or 1
    model, phys = cpu_info()
    n_blk = 64
    I, Q = make_iq(64, n_blk, fc=6290.0, A=0.25)
    t1, _ = ao.bench_run(0, I, Q, 1)                       # calibrate: 1 thread
    rate1 = I.size / t1
    # probe the all-thread rate on a short sample, then size the real one for ~ seconds_budget / 3 (>= 2 s)
    n_probe = max(threads, 256)
    Ip, Qp = make_iq(n_probe, n_blk, fc=6290.0, A=0.25)
    tp, _ = ao.bench_run(0, Ip, Qp, threads)
    rate_p = Ip.size / tp
    target_s = max(2.0, seconds_budget / 3.0)
    n_ch = int(min(65536, max(threads, rate_p * target_s / (n_blk * BLOCK))))   # <= 1 GiB of int16 per input array
    n_ch -= n_ch % threads
    n_ch = max(n_ch, threads)
    base = make_iq(min(n_ch, 2048), n_blk, fc=6290.0, A=0.25)
    import numpy as np
    reps = (n_ch + base[0].shape[0] - 1) // base[0].shape[0]
    I = np.tile(base[0], (reps, 1, 1))[:n_ch]; Q = np.tile(base[1], (reps, 1, 1))[:n_ch]
    tn, _ = ao.bench_run(0, I, Q, threads)
    rate_n = I.size / tn
    cores = phys or threads
    # how many threads' worth of plain arithmetic the box gives this process on `threads` threads (sandboxes expose more
    # logical CPUs than they schedule): the yardstick for the all-thread figure, measured, not assumed
    capacity = ao.host_parallel_capacity(threads)
    scaling = rate_n / (rate1 * cores)
    note = ""
    if scaling < 0.5:
        note = ("; all-thread rate = %.2f x (1-thread rate x %d physical cores) because this host schedules only %.1f threads' "
                "worth of arithmetic for %d threads of one process (register-only float loop, ao_spin_calibrate): "
                "against that capacity the oracle scales %.2f x" % (scaling, cores, capacity, threads, rate_n / (rate1 * capacity)))
    return {"value": round(rate_n / 1e6, 3), "unit": "Msamples/s", "cores": threads, "physical_cores": phys, "cpu_model": model,
            "kind": "port", "one_thread_value": round(rate1 / 1e6, 3), "sample_seconds": round(tn, 3),
            "host_parallel_capacity_threads": round(capacity, 2),
            "sample": "CPU oracle (oracle/asdr_oracle.c, gcc -O2 -ffp-contract=off), C2 USB chain, %d channels x %d blocks on %d "
                      "threads, update() only in the timed region (instances created and configured before it)%s"
                      % (n_ch, n_blk, threads, note)}


def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def self_launch(args):
    """--gpus N > 1 without a launcher: start N ranks as a child job.  Nothing in THIS process has touched torch or HIP."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)   # 2000 x ~0.14 ms: long enough for the clocks to settle (the first ~500 launches after idle run ~3 % slower)
    ap.add_argument("--warmup", type=int, default=500)
    ap.add_argument("--channels", type=int, default=CHANNELS_PER_GPU, help="channels per GPU (default = C2)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dry-run", action="store_true", help="no GPU: gloo, empty timed loop (tests the launch plumbing)")
    args = ap.parse_args()

    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if not launched and args.gpus > 1:
        sys.exit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1")) if launched else 1
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but the launcher started %d ranks (WORLD_SIZE); refusing to print a mislabelled line\n"
                         % (args.gpus, world))
        sys.exit(2)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    import numpy as np
    import torch
    dist = None
    if launched:   # one process per GPU; RCCL only for the barrier and the max-over-ranks time
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.dry_run:
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    elif not args.dry_run:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cpu") if args.dry_run else torch.device("cuda", local_rank)
    n_ch = args.channels

    batch = None
    if not args.dry_run:
        import audiosdr_amd as A
        from audiosdr_amd.synth import make_iq
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
        if batch is not None:
            b = i % N_INPUT_BLOCKS
            batch.update_device(dI[b].data_ptr(), dQ[b].data_ptr(), dOut.data_ptr(), 1, stream)

    def fence():
        if not args.dry_run:
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        if not args.dry_run:
            torch.cuda.synchronize()

    # The GPU's clocks need ~0.2 s of work to settle after idle (the first ~500 launches run ~3 % slower, a 20-step run right
    # after start-up ~10 %): whatever --warmup says, at least SETTLE_LAUNCHES untimed launches precede the timed region.  The
    # timed region is exactly --steps launches.
    settle = max(0, SETTLE_LAUNCHES - args.warmup) if batch is not None else 0
    for i in range(settle):
        step(i)
    for i in range(args.warmup):
        step(i)
    fence()
    # Kernel time: ONE HIP-event pair, recorded by the library on the launch stream, around the K launches of the timed region;
    # kernel_ms = elapsed / K = the average time per back-to-back launch, gaps between the kernels included.  (A pair around
    # every launch puts two more packets between consecutive kernels: it slows the job by ~5 % and measures its own overhead;
    # that per-launch figure is taken AFTER the timed region, from 40 further launches, and reported beside it.)
    fence()
    if batch is not None:
        batch.region_timing_begin(stream)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    if batch is not None:
        region_ms, region_calls = batch.region_timing_end()     # records the stop event, synchronises the stream
    fence()
    t1 = time.perf_counter()
    wall = t1 - t0
    kernel_ms = [region_ms / max(1, region_calls)] if batch is not None else [0.0]
    pair_ms = None
    if batch is not None:
        batch.kernel_timing_begin(40)
        for i in range(40):
            step(i)
        pair_ms = float(np.mean(batch.kernel_timing_end(40)))
    if dist is not None:
        tw = torch.tensor([wall], dtype=torch.float64, device=dev)
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        wall = float(tw.item())

    if rank == 0:
        samples = float(world) * n_ch * BLOCK * args.steps
        k_ms = float(np.mean(kernel_ms))
        ach = ALGO_BYTES_PER_BLOCK * n_ch / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        # HBM bytes per launch: NOT measured in this run -- copied from the committed rocprofv3 PMC passes of this same command
        # (tools/prof_pmc.sh); dropped when that file was taken from a different build of the library
        traffic, traffic_source = None, None
        try:
            with open(os.path.join(ROOT, PMC_FILE)) as f:
                pj = json.load(f)
            if n_ch == CHANNELS_PER_GPU and not args.dry_run:
                import audiosdr_amd as A
                from audiosdr_amd import build as _build
                same_binary = pj.get("library_sha256") == A.library_sha256()
                same_source = pj.get("source_sha256") is not None and pj.get("source_sha256") == _build.source_sha256()
                if same_binary or same_source:   # (the binary's hash depends on the tree's path; the sources' does not)
                    traffic = pj.get("traffic_bytes_per_launch")
                    traffic_source = "%s (builder-run rocprofv3 --pmc passes of this command on a library built from these sources: %s)" % (
                        PMC_FILE, ("binary sha256 %s" % pj.get("library_sha256", "")[:12]) if same_binary else ("source sha256 %s" % pj.get("source_sha256", "")[:12]))
                else:
                    traffic_source = "%s is from another build of the library: dropped" % PMC_FILE
        except Exception:
            pass
        out = {
            "metric": "Msamples/s through full SSB demod chain, batched 128-sample blocks",
            "value": round(samples / wall / 1e6, 2), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(wall / args.steps * 1e3, 5), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "C2: SSB (USB) demod, %d channels/GPU x 1 block/step, NB+IF+mixer+Hilbert+audio IIR+AGC" % n_ch,
                       "untimed_launches_before_the_timed_region": settle + args.warmup,
                       "channels_per_gpu": n_ch, "block": BLOCK, "sharding": "channels, no collective",
                       "input": "tone + LCG noise per channel, no impulses: the blanker runs (envelope, average, threshold) but detects "
                                "nothing, so its mask stays all ones; with an impulse in every block of every channel the same chain "
                                "is ~9 % slower (tools/bench_configs.py c2)"},
            "roofline": {"bound": "hbm", "kernel": "asdr_update_kernel", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                         "kernel_ms": round(k_ms, 5),
                         "kernel_ms_method": "one HIP-event pair on the launch stream around the %d timed launches / %d" % (args.steps, args.steps),
                         "kernel_ms_event_pair_per_launch": None if pair_ms is None else round(pair_ms, 5),
                         "algorithmic_bytes_per_launch": ALGO_BYTES_PER_BLOCK * n_ch,
                         "hbm_read_share_frac": round(ALGO_READ_BYTES_PER_BLOCK * n_ch / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if k_ms > 0 else 0.0},
        }
        if args.dry_run:
            out["dry_run"] = True
        if world == 1 and not args.no_cpu_baseline and not args.dry_run:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if batch is not None:
        batch.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
