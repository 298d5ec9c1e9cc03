#!/usr/bin/env python3
"""bench.py -- Msamples/s through the full SSB demod chain, batched 128-sample blocks (BASELINE.json).

Default workload (`--config c2`, weak scaling: channels shard embarrassingly, no collective):
  SURVEY.md 8d config C2 per GPU: 65,536 independent channels x one 128-sample block per step,
  USB demodulation, noise blanker on (default threshold 1.2), IF band-pass, complex mixer,
  257-tap Hilbert, audio IIR filter enabled (bw2700), AGC default, ALS off.
  int16 I/Q rows and the int16 audio rows are resident in HBM before the timed region.
A "step" is one asdr_update_device() call = one pass of the hot path over the whole batch.  The calls go to ASDR_STREAM_BATCH
(include/asdr.h): the batch's own streams, ordered against the batch's other calls only -- a batch of one settings group then runs
its two halves as two never-joined lanes, so that one launch's tail is filled by the next launch of the other half.  The same
steps on a caller's stream (every call ordered behind the previous one as a whole) are timed right after the headline region and
reported as `roofline.caller_stream_ordered` (`--caller-stream` makes that the headline instead).

The other two BASELINE configs that need more than one GPU are selectable, so that the driver's one command covers them
(strong scaling: the job is fixed, --gpus N ranks each take 1/N of its channels):
  --config c4   1,048,576 channels in all, mode = channel mod 7 (LSB, USB, CW_LSB, CW_USB, AM, SAM, WSPR), ALS notch on,
                blanker at 10 dB; one block per step
  --config c5   4,096 WSPR receivers in all (EXTRAS/BareBonesWSPR/BareBonesWSPR.ino:87-102 settings); one step = one
                646-block call (1/64 of a 2-minute slot) appended to the capture sink

Launching:
  python bench.py --gpus N ...            N > 1 without a distributed launcher: this process starts
                                          `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD
                                          (before anything here touches torch or the GPU) and relays rank 0's JSON line
  python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...   (what the driver does for N > 1)
  A launcher whose WORLD_SIZE differs from --gpus is an error (exit 2), never a silently mislabelled line.
  --dry-run: no GPU and no HIP library: gloo backend, the timed loop is empty; checks the launch / reduce / JSON plumbing.

Warm-up: `--warmup W` untimed steps of the measured batch precede the K timed steps, exactly as asked (`warmup` in the JSON = W).
The GPU's clocks need ~0.2 s of work to leave their idle state (a 20-step run right after start-up is ~10 % slow), so BEFORE the
measured batch exists a scratch batch of the same configuration runs `--settle S` launches (default 1500 for c2 / c4, 4 for c5) and
is destroyed: device pre-heating, not steps of the job -- reported as `config.clock_settle_launches_on_a_scratch_batch`; `--settle 0`
switches it off.  Without `--warmup` the default is W = 20 (c5: 2).

`h2d_d2h_inclusive` (c2, N = 1): the same job through the HOST-pointer entry point asdr_update(), the boundary the reference's own
data path has (host-resident audio blocks: AudioSDR.cpp:46-47, 158-167) -- 768 bytes per channel-block cross PCIe, overlapped in
channel-range chunks (H2D || kernels || D2H); measured after the headline region with the host's clock around synchronous calls on
pinned caller buffers, with pageable buffers and with the overlap switched off beside it.  Never `value`.

`--single-process` (with --gpus N): ONE process drives N GPUs through a sharded batch (asdr_create_sharded: shard g = channels
[g*C/N, (g+1)*C/N) on GPU g, no collective), device-resident rows per shard, asynchronous calls on every shard's stream; the line
says so in `config.launch`.  `--devices 0,0` puts the shards on the listed ordinals (several on one GPU: the test on a 1-GPU box).

Prints ONE JSON line (rank 0).  `value` = samples processed by all ranks / max-over-ranks wall time.
`roofline` is for the dominant kernel of the config: algorithmic bytes per step / the mean duration of a step
from ONE HIP-event pair recorded on the launch stream around the timed region.  `robustness` (c2, N = 1): the same 65,536-channel
chain on adverse inputs / settings that defeat its data-dependent fast paths, measured the same way after the headline region.
`cpu_baseline` is the CPU oracle (a port, the reference is unbuildable here) timed on this host on a bounded sample of C2.
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

BLOCK = 128
HBM_PEAK_GBS = 8000.0                 # MI355X_MICROARCH.md: 8.0 TB/s spec
N_INPUT_BLOCKS = 4                    # distinct resident input blocks cycled through by the steps
PMC_FILE = os.path.join("profiles", "pmc_latest.json")
ISSUE_FILE = os.path.join("profiles", "issue_latest.json")
LATENCY_FILE = os.path.join("profiles", "roofline_latency_latest.json")

# ---- algorithmic bytes per channel-block, SURVEY.md 8d (minimal carried state read AND written once per block at T = 1) ----
B_IO, B_PARAMS = 768, 96
ST_NB, ST_IF, ST_PHASE, ST_HILBERT, ST_IMG, ST_PLL, ST_AF, ST_AGC, ST_ALS = 3076, 128, 4, 1536, 128 + 4 + 4, 40, 64, 20, 450
C2_STATE = ST_NB + ST_IF + ST_PHASE + ST_HILBERT + ST_AF + ST_AGC                         # 4828
ALGO_BYTES_PER_BLOCK = B_IO + B_PARAMS + 2 * C2_STATE                                    # 10,520
ALGO_READ_BYTES_PER_BLOCK = 512 + B_PARAMS + C2_STATE                                    # 5,436: the HBM-read share
_C4_SSB = ST_NB + ST_IF + ST_PHASE + ST_HILBERT + ST_AGC + ST_ALS                        # 5 of 7 channels
_C4_AM = ST_NB + ST_IF + ST_IMG + ST_AGC + ST_ALS
_C4_SAM = ST_NB + ST_IF + ST_PLL + ST_IMG + ST_AGC + ST_ALS
C4_ALGO_BYTES_PER_BLOCK = B_IO + B_PARAMS + 2.0 * (5 * _C4_SSB + _C4_AM + _C4_SAM) / 7.0  # 10,501.1 on average
C5_T = 646                                                                               # blocks per call: 64 calls = one 2-minute slot
C5_STATE = ST_IF + ST_PHASE + ST_HILBERT + ST_AGC                                        # 1688 (blanker, audio filter off)
C5_ALGO_BYTES_PER_BLOCK = (B_IO * C5_T + B_PARAMS + 2.0 * C5_STATE) / C5_T                # 773.4

CONFIGS = {
    # name: (channels of the whole job or per GPU, scaling, blocks per step, default settle steps)
    "c2": dict(channels=65536, per_gpu=True, scaling="weak", blocks=1, settle=1500, algo=ALGO_BYTES_PER_BLOCK,
               kernel="asdr_update_kernel_mw"),   # (the label comes from the library's launch census of the timed region: dominant_kernel; this is the dry-run fallback)
    "c4": dict(channels=1048576, per_gpu=False, scaling="strong", blocks=1, settle=300, algo=C4_ALGO_BYTES_PER_BLOCK,
               kernel="asdr_update_kernel_als_small_one (+ the SAM pre | PLL | post launches and the remainders' launch beside it)"),
    "c5": dict(channels=4096, per_gpu=False, scaling="strong", blocks=C5_T, settle=4, algo=C5_ALGO_BYTES_PER_BLOCK,
               kernel="asdr_stream_kernel"),
}
CHANNELS_PER_GPU = CONFIGS["c2"]["channels"]


def dominant_kernel(launched, fallback):
    """The kernel the roofline object is about: of the instantiations the library launched in the timed region (its launch census), the one the
    region's time belongs to -- the chain kernel with the most launches (the PLL / role / snapshot launches beside it are named in the census)."""
    chain = {k: n for k, n in launched.items() if k.startswith(("asdr_update_kernel", "asdr_stream_kernel"))}
    if not chain:
        return fallback if not launched else max(launched, key=launched.get)
    return max(chain, key=chain.get)


def configure_c2(sdr):
    sdr.setDemodMode(1)        # USBmode
    sdr.enableAudioFilter()    # bw2700 from init(); NB + AGC are on by default


def configure_c4(sdr, lib=None, channel0=0):
    """mode = (global channel index) mod 7, ALS notch (defaults M 55 / lambda 0.5 / delay 3), blanker threshold 10 dB."""
    n = sdr.n_channels
    if lib is not None:
        for c in range(n):
            lib.asdr_setDemodMode(sdr._h, c, (channel0 + c) % 7)
    else:
        for c in range(n):
            sdr.setDemodMode((channel0 + c) % 7, ch=c)
    sdr.enableALSfilter(); sdr.setNoiseBlankerThresholdDb(10.0)


def configure_c5(sdr):   # BareBonesWSPR.ino:87-102,129
    sdr.enableAGC(); sdr.setAGCmode(2); sdr.disableALSfilter(); sdr.disableNoiseBlanker(); sdr.setNoiseBlankerThresholdDb(10.0)
    sdr.setInputGain(1.0); sdr.setOutputGain(0.5); sdr.setIQgainBalance(1.020); sdr.setAudioFilter(2); sdr.setDemodMode(6); sdr.setMute(0)


# ---- the adverse cases of the `robustness` object (each defeats one of the C2 kernel's data- / settings-dependent fast paths) ----
ROBUSTNESS_CASES = {
    "divergent_mixer_phases": dict(phases=True, agc=False, impulses=False,
        what="every wave holds 8 channels with 8 DIFFERENT mixer phases (channel c switched LSB -> USB after c mod 8 blocks): the "
             "local-oscillator cache and the wave-uniform mixer both miss; every lane runs its own 2 x 16 sin/cos lookups"),
    "agc_general_form": dict(phases=False, agc=True, impulses=False,
        what="AGC hang time 0 (setAGChangTime(0)): the envelope releases between the audio's peaks and attacks again at each of them, "
             "so every 8-sample chunk takes the AGC's general per-sample form (no quiet-block, hanging-chunk or attack-only fast form); input as C2"),
    "impulse_every_block": dict(phases=False, agc=False, impulses=True,
        what="an impulse in every block of every channel: the blanker's general path (counts, mask decode / zero / ramp / encode)"),
    "all_three": dict(phases=True, agc=True, impulses=True, what="the three together"),
}


def stagger_divergent_phases(batch, step_one_block, lib=None, oracles=None):
    """Leaves the batch in USB mode with 8 different mixer phases in every wave: all channels start in LSB; before block j
    (j = 0..7) the channels with c mod 8 == j switch to USB (AudioSDR.cpp:187-222: the mixer phase is kept, the increment
    changes), so channel c has accumulated (c mod 8) blocks of the LSB increment.  `step_one_block(j)` runs block j."""
    n = batch.n_channels
    batch.setDemodMode(0)
    for o in (oracles or []):
        o.setDemodMode(0)
    for j in range(8):
        if lib is not None:
            for c in range(j, n, 8):
                lib.asdr_setDemodMode(batch._h, c, 1)
        else:
            for c in range(j, n, 8):
                batch.setDemodMode(1, ch=c)
        for c, o in enumerate(oracles or []):
            if c % 8 == j:
                o.setDemodMode(1)
        step_one_block(j)


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
            "kind": "port", "one_thread_value": round(rate1 / 1e6, 3),
            "one_thread_value_times_physical_cores": round(rate1 * cores / 1e6, 1),   # the scale to read the GPU figure against: what this host's cores would give if the box scheduled them all
            "sample_seconds": round(tn, 3),
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


def tiled_input(np, torch, dev, n_ch, n_blocks, uniq, channel0=0, per_block=True, **sig):
    """`uniq` distinct channels generated (audiosdr_amd/synth.py) and tiled over n_ch; resident tensors, one [n_ch][128] pair per
    block (per_block) or one [n_ch][n_blocks][128] pair."""
    from audiosdr_amd.synth import make_iq
    uniq = max(8, min(uniq, n_ch))
    I, Q = make_iq(uniq, n_blocks, channel0=channel0, **sig)
    reps = (n_ch + uniq - 1) // uniq
    if per_block:
        dI = [torch.from_numpy(np.ascontiguousarray(I[:, b])).to(dev).repeat(reps, 1)[:n_ch].contiguous() for b in range(n_blocks)]
        dQ = [torch.from_numpy(np.ascontiguousarray(Q[:, b])).to(dev).repeat(reps, 1)[:n_ch].contiguous() for b in range(n_blocks)]
        return dI, dQ
    dI = torch.from_numpy(I).to(dev).repeat(reps, 1, 1)[:n_ch].contiguous()
    dQ = torch.from_numpy(Q).to(dev).repeat(reps, 1, 1)[:n_ch].contiguous()
    return dI, dQ


def measure_region(batch, step, stream, warm, timed):
    """`warm` untimed steps, then ONE HIP-event pair (recorded by the library on the launch stream) around `timed` steps.
    Returns milliseconds per step."""
    import torch
    for i in range(warm):
        step(i)
    torch.cuda.synchronize()
    batch.region_timing_begin(stream)
    for i in range(timed):
        step(i)
    ms, calls = batch.region_timing_end()
    return ms / max(1, calls)


def robustness(np, torch, dev, local_rank, n_ch, stream, timed):
    """The C2 chain (65,536 channels, USB + blanker + audio filter + AGC) on the adverse cases of ROBUSTNESS_CASES: same kernel,
    same event-pair method as the headline region, >= 200 timed launches each after the case's own warm-up."""
    import audiosdr_amd as A
    L = A.load_library()
    out = {}
    dOut = torch.empty((n_ch, BLOCK), dtype=torch.int16, device=dev)
    for name, c in ROBUSTNESS_CASES.items():
        sig = dict(fc=6290.0, A=0.25)
        if c["impulses"]:
            sig.update(impulse_every=128)     # sample 64 of every block
        dI, dQ = tiled_input(np, torch, dev, n_ch, N_INPUT_BLOCKS, n_ch // 4, **sig)
        torch.cuda.synchronize()   # (torch's stream has produced the inputs before the batch's own streams read them)
        batch = A.AudioSDRBatch(n_ch, device=local_rank)

        def step(i, batch=batch, dI=dI, dQ=dQ):
            b = i % N_INPUT_BLOCKS
            batch.update_device(dI[b].data_ptr(), dQ[b].data_ptr(), dOut.data_ptr(), 1, stream)

        if c["phases"]:
            stagger_divergent_phases(batch, step, lib=L)
        configure_c2(batch)
        if c["agc"]:
            batch.setAGChangTime(0.0)
        ms = measure_region(batch, step, stream, 120, timed)   # 120 blocks: the blanker's average and the AGC have settled
        st = batch.read_status()
        ach = ALGO_BYTES_PER_BLOCK * n_ch / (ms * 1e-3) / 1e9
        out[name] = {"kernel_ms": round(ms, 5), "frac": round(ach / HBM_PEAK_GBS, 4), "Msamples_per_s": round(n_ch * BLOCK / ms / 1e3, 1),
                     "launches_timed": timed, "channels_with_blanker_detection": int(st["nb_detected"].sum()), "what": c["what"]}
        batch.close()
        del dI, dQ
    return out


C3_STATE = ST_NB + ST_IF + ST_PLL + ST_IMG + ST_AF + ST_AGC                               # 3464
C3_ALGO_BYTES_PER_BLOCK = B_IO + B_PARAMS + 2 * C3_STATE                                 # 7,792


def configure_c3(sdr):   # SAM, blanker at 10 dB (else no lock, Q2), audio filter audioAM, AGC default
    sdr.setDemodMode(5); sdr.setNoiseBlankerThresholdDb(10.0); sdr.enableAudioFilter(); sdr.setAudioFilter(0)


def other_configs(np, torch, dev, local_rank):
    """BASELINE configs 3, 4 and 5 (their single-GPU shares) ON THE DRIVER'S RECORD: the default `bench.py --gpus 1` line carries, beside
    C2's headline, one short measured region per config -- same method (one HIP-event pair on the launch stream, ASDR_STREAM_BATCH:
    lanes where the schedule has them), SURVEY 8(d)'s bytes, and a one-channel bit-for-bit spot check against the CPU oracle of the
    blocks in front of the region.  C3 and C4 carry TWO regions, like the C2 line (its timed window | `roofline.steady_state`):
    `fresh_bank_ms_per_step` = the steps right behind the spot check (a fresh bank: AGC attacking, PLLs settling) and `ms_per_step` /
    `frac` = the same region 300 steps later.  Sized to add a few seconds to the command."""
    import audiosdr_amd as A
    from audiosdr_amd.synth import make_iq
    from oracle import asdr_oracle as ao
    L = A.load_library()
    stream = A.STREAM_BATCH
    res = {}

    def run(name, n_ch, T, n_in, uniq, sig, configure_batch, configure_oracle, check_ch, warm, timed, algo, kernel, what, capture=False, settle=0):
        uniq = min(uniq, n_ch)
        I, Q = make_iq(uniq, n_in * T, **sig)                       # [uniq][n_in * T][128]
        reps = (n_ch + uniq - 1) // uniq
        t_in = [(torch.from_numpy(np.ascontiguousarray(I[:, b * T:(b + 1) * T])).to(dev).repeat(reps, 1, 1)[:n_ch].contiguous(),
                 torch.from_numpy(np.ascontiguousarray(Q[:, b * T:(b + 1) * T])).to(dev).repeat(reps, 1, 1)[:n_ch].contiguous()) for b in range(n_in)]
        dOut = None if capture else torch.empty((n_ch, T, BLOCK), dtype=torch.int16, device=dev)
        torch.cuda.synchronize()                                    # (the inputs are torch's work on torch's stream: done before the batch's own streams read them)
        batch = A.AudioSDRBatch(n_ch, device=local_rank)
        configure_batch(batch)
        if capture:
            batch.capture_open(T * (warm + 1))

        def step(i):
            dI, dQ = t_in[i % n_in]
            if capture:
                if batch.capture_position + T > batch.capture_capacity:
                    batch.capture_rewind()
                batch.capture_update_device(dI.data_ptr(), dQ.data_ptr(), T, None, stream)
            else:
                batch.update_device(dI.data_ptr(), dQ.data_ptr(), dOut.data_ptr(), T, stream)

        lc0 = batch.lane_calls()
        for i in range(warm):
            step(i)
        batch.synchronize(); torch.cuda.synchronize()
        # spot check: channel check_ch after the `warm` calls, against the oracle fed the same blocks
        o = ao.OracleSDR(); configure_oracle(o, check_ch)
        cu = check_ch % uniq
        xi = np.concatenate([I[cu, (i % n_in) * T:(i % n_in + 1) * T] for i in range(warm)])
        xq = np.concatenate([Q[cu, (i % n_in) * T:(i % n_in + 1) * T] for i in range(warm)])
        want = o.update(xi, xq).reshape(-1, BLOCK)
        if capture:
            got = batch.capture_read(check_ch).reshape(-1, BLOCK)
            ok = bool(np.array_equal(got, want[:got.shape[0]])) and got.shape[0] == want.shape[0]
        else:
            got = dOut[check_ch].cpu().numpy().reshape(-1, BLOCK)
            ok = bool(np.array_equal(got, want[-T:]))
        extra = {}
        if name == "c3":
            extra["lock_fraction"] = float(batch.read_status()["sam_locked"].mean())
        if capture:
            batch.capture_rewind()
        A.binding.kernels_launched(reset=True)   # (the launch census names the instantiations the measured region ran)
        ms_fresh = measure_region(batch, step, stream, 0, timed)   # blocks warm .. warm + timed - 1 of a FRESH bank (AGC attacking, locks settling)
        ms = ms_fresh
        if settle > 0:   # ... and the same region again once the bank has run `settle` more steps (what the C2 line calls its steady state)
            for i in range(settle):
                step(i)
            A.binding.kernels_launched(reset=True)
            ms = measure_region(batch, step, stream, 0, timed)
        launched = A.binding.kernels_launched()
        ach = algo * n_ch * T / (ms * 1e-3) / 1e9
        d = {"workload": what, "channels": n_ch, "blocks_per_step": T, "ms_per_step": round(ms, 5), "Msamples_per_s": round(n_ch * T * BLOCK / ms / 1e3, 1),
             "frac": round(ach / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_channel_block": round(algo, 1), "dominant_kernel": kernel,
             "steps_timed": timed, "steps_untimed": warm + (timed + settle if settle > 0 else 0), "fresh_bank_ms_per_step": round(ms_fresh, 5),
             "fresh_bank_region": "steps %d..%d of the bank" % (warm, warm + timed - 1), "lane_calls": batch.lane_calls() - lc0,
             "oracle_spot_check": {"channel": check_ch, "blocks": int(want.shape[0]) if capture else T, "bit_exact": ok},
             "kernels_launched_in_the_measured_region": launched}
        if capture:
            d["times_real_time"] = round(T * BLOCK / 44100.0 / (ms * 1e-3), 1)
            d["calls_run_as_block_pipeline"] = batch.stream_pipeline_launches()
        d.update(extra)
        batch.close()
        del t_in, dOut
        res[name] = d

    uniq = 3584
    fc3 = 6890.0 + (np.arange(uniq) % 7 - 3) * 50.0
    run("c3", 262144, 1, 16, uniq, dict(fc=fc3, A=0.3, m=0.5, fm=400.0), configure_c3, lambda o, c: configure_c3(o), 262143, 16, 96,
        C3_ALGO_BYTES_PER_BLOCK, "asdr_sam_pre_kernel_uniform | asdr_sam_pll_kernel | asdr_sam_post_kernel_uniform (three launches per block)",
        "C3: SAM + PLL carrier lock + AGC, 262,144 channels x 1 block/step; carriers 6890 + (c mod 7 - 3) x 50 Hz, 50 % AM at 400 Hz", settle=300)

    def c4_oracle(o, c):
        o.setDemodMode(c % 7); o.enableALSfilter(); o.setNoiseBlankerThresholdDb(10.0)
    run("c4_share", 131072, 1, 12, uniq, dict(fc=6890.0 - 300, A=0.3, m=0.4, f2=7500.0, a2=0.15), lambda b: configure_c4(b, lib=L), c4_oracle, 100001, 12, 96,
        C4_ALGO_BYTES_PER_BLOCK, "asdr_update_kernel_als_small_one (+ the SAM pre | PLL | post-with-ALS launches and the remainders' launch beside it)",
        "C4: one GPU's share (131,072 of 1,048,576 channels) of the mixed-mode batch: mode = channel mod 7, ALS notch, blanker at 10 dB; 1 block/step", settle=300)
    run("c5_share", 512, C5_T, 1, 512, dict(fc=6890.0, A=0.02, noise=0.05), configure_c5, lambda o, c: configure_c5(o), 511, 2, 8,
        C5_ALGO_BYTES_PER_BLOCK, "asdr_stream_kernel_h3 (block pipeline, three FIR helper waves: the launch census beside this says which form ran)",
        "C5: one GPU's share (512 of 4,096 WSPR receivers, BareBonesWSPR.ino settings); one step = a 646-block call into the capture sink", capture=True)
    return res


PCIE_REF_GBS = 63.0   # PCIe 5.0 x16, one direction (the figure the round-3 review prices the 768 B per channel-block against)


def host_path(np, local_rank, n_ch, calls=30, only_autopin=False):
    """The C2 job through asdr_update() (host pointers): Msamples/s with the host's clock around `calls` synchronous calls, for pinned
    caller buffers (the figure), pageable caller buffers (staged through the batch's pinned area by worker threads) and with the
    overlap switched off (one chunk: H2D -> kernels -> D2H serially, round 3's behaviour)."""
    import audiosdr_amd as A
    from audiosdr_amd.synth import make_iq
    uniq = max(8, n_ch // 16)
    bI, bQ = make_iq(uniq, N_INPUT_BLOCKS, fc=6290.0, A=0.25)
    reps = (n_ch + uniq - 1) // uniq
    res = {}
    pin = [A.host_alloc((n_ch, 1, BLOCK)) for _ in range(2 * N_INPUT_BLOCKS + 1)]
    pag = [np.empty((n_ch, 1, BLOCK), np.int16) for _ in range(2 * N_INPUT_BLOCKS + 1)]
    for bufs in (pin, pag):
        for b in range(N_INPUT_BLOCKS):
            bufs[2 * b][:] = np.tile(bI[:, b:b + 1], (reps, 1, 1))[:n_ch]
            bufs[2 * b + 1][:] = np.tile(bQ[:, b:b + 1], (reps, 1, 1))[:n_ch]
    legs = [("pinned", pin, 0, calls), ("pageable", pag, 0, max(6, calls // 3)), ("pinned_no_overlap", pin, 1, max(6, calls // 3))]
    if only_autopin:
        legs = [("pageable_autopin", pag, 0, max(6, calls // 3))]
    for name, bufs, chunks, n_calls in legs:
        batch = A.AudioSDRBatch(n_ch, device=local_rank)
        configure_c2(batch)
        batch.set_host_chunks(chunks)
        A.binding.host_autopin(1 if name == "pageable_autopin" else 0)   # (opt-in: recurring pageable ranges are registered in place at their second call)
        for i in range(4):
            batch.update_into(bufs[2 * (i % N_INPUT_BLOCKS)], bufs[2 * (i % N_INPUT_BLOCKS) + 1], bufs[-1])
        t0 = time.perf_counter()
        for i in range(n_calls):
            batch.update_into(bufs[2 * (i % N_INPUT_BLOCKS)], bufs[2 * (i % N_INPUT_BLOCKS) + 1], bufs[-1])
        dt = (time.perf_counter() - t0) / n_calls
        info = batch.host_path_info()
        res[name] = {"ms_per_call": round(dt * 1e3, 4), "Msamples_per_s": round(n_ch * BLOCK / dt / 1e6, 1),
                     "pcie_GBps": round(768.0 * n_ch / dt / 1e9, 2), "chunks": info["chunks"], "calls_timed": n_calls}
        batch.close()
        A.binding.host_autopin_clear(); A.binding.host_autopin(0)
    for a in pin:
        A.host_free(a)
    if only_autopin:
        return res["pageable_autopin"]
    # The opt-in leg runs in a CHILD process: it registers ordinary numpy memory with the runtime (hipHostRegister) and releases it again, and a
    # long-lived process that had done so was seen to abort in a later, unrelated copy (tests/test_gpu_host_path.py `isolated`) -- not in this one.
    res["pageable_autopin"] = None
    try:
        import subprocess
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--autopin-leg", "--channels", str(n_ch), "--device", str(local_rank)],
                           stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=300)
        if r.returncode == 0:
            res["pageable_autopin"] = json.loads(r.stdout.strip().splitlines()[-1])
    except Exception:   # noqa: BLE001  (the leg is an extra: its failure must not take the bench line with it)
        pass
    p = res["pinned"]
    return {"value": p["Msamples_per_s"], "unit": "Msamples/s", "ms_per_call": p["ms_per_call"], "pcie_GBps": p["pcie_GBps"],
            "frac_of_63GBps": round(p["pcie_GBps"] / PCIE_REF_GBS, 3), "chunks": p["chunks"],
            "bytes_per_channel_block": 768, "pageable": res["pageable"], "pageable_autopin": res["pageable_autopin"], "no_overlap": res["pinned_no_overlap"],
            "method": "asdr_update() on %d channels x 1 block per call, caller buffers in page-locked host memory (asdr_host_alloc), "
                      "host clock around %d synchronous calls after 4 untimed ones; pcie_GBps = 768 B x channels / time (both directions "
                      "summed: 512 in + 256 out, which PCIe moves concurrently); `pageable`: ordinary numpy buffers staged by the "
                      "library's copy threads (the default for memory the caller did not pin); `pageable_autopin`: the same buffers with asdr_host_autopin(1) "
                      "-- registered in place at their second call, opt-in because the caller must then not free them behind the library's back; `no_overlap`: asdr_set_host_chunks(1)" % (n_ch, calls)}


def main_single_process(args, cfg, settle_min, warm_req):
    """One process, N GPUs: a sharded batch (asdr_create_sharded), per-shard device-resident rows, asynchronous calls on every shard."""
    import numpy as np
    import torch
    import audiosdr_amd as A
    devices = [int(d) for d in args.devices.split(",")] if args.devices else list(range(args.gpus))
    G = len(devices)
    if G != args.gpus:
        sys.stderr.write("bench.py: --devices lists %d ordinals for --gpus %d\n" % (G, args.gpus)); sys.exit(2)
    if args.config == "c5":
        sys.stderr.write("bench.py: --single-process covers c2 and c4\n"); sys.exit(2)
    total_ch = (args.channels or cfg["channels"]) * (G if (cfg["per_gpu"] or args.channels) else 1)
    batch = A.AudioSDRBatch(total_ch, devices=devices)
    if args.config == "c2":
        configure_c2(batch)
    else:
        configure_c4(batch, lib=A.load_library(), channel0=0)
    shards, bufs = [], []
    for g in range(G):
        lo, hi = batch.shard_range(g)
        dev = torch.device("cuda", devices[g])
        with torch.cuda.device(dev):
            if args.config == "c2":
                dI, dQ = tiled_input(np, torch, dev, hi - lo, N_INPUT_BLOCKS, (hi - lo) // 4, channel0=lo, fc=6290.0, A=0.25)
            else:
                dI, dQ = tiled_input(np, torch, dev, hi - lo, N_INPUT_BLOCKS, 3584, channel0=lo % 3584, fc=6890.0 - 300, A=0.3, m=0.4, f2=7500.0, a2=0.15)
            dOut = torch.empty((hi - lo, BLOCK), dtype=torch.int16, device=dev)
            stream = torch.cuda.Stream(device=dev)
        shards.append(batch.shard(g)); bufs.append((dI, dQ, dOut, stream))
    for d in sorted(set(devices)):
        torch.cuda.synchronize(d)   # the inputs (torch's default streams) are complete before the shards' streams read them

    def step(i):
        b = i % N_INPUT_BLOCKS
        for v, (dI, dQ, dOut, st) in zip(shards, bufs):
            v.update_device(dI[b].data_ptr(), dQ[b].data_ptr(), dOut.data_ptr(), 1, st.cuda_stream)

    def sync():
        for d in sorted(set(devices)):
            torch.cuda.synchronize(d)

    for i in range(max(settle_min, 0) + args.warmup):
        step(i)
    sync()
    A.binding.kernels_launched(reset=True)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    sync()
    wall = time.perf_counter() - t0
    launched = A.binding.kernels_launched()
    per_gpu_ms = wall / args.steps * 1e3
    algo = cfg["algo"] * total_ch
    ach = algo / G / (per_gpu_ms * 1e-3) / 1e9      # per GPU: each holds 1/G of the channels
    out = {"metric": "Msamples/s through full SSB demod chain, batched 128-sample blocks",
           "value": round(float(total_ch) * BLOCK * args.steps / wall / 1e6, 2), "unit": "Msamples/s", "n_gpus": G, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": round(per_gpu_ms, 5), "higher_is_better": True, "scaling": cfg["scaling"],
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "%s through ONE process and a sharded batch: %d channels in all, shard g on device %s" % (args.config, total_ch, devices),
                      "name": args.config, "launch": "single process, asdr_create_sharded, asynchronous asdr_update_device on every shard's stream",
                      "devices": devices, "warmup_requested": warm_req, "clock_settle_launches": settle_min, "channels_total": total_ch,
                      "sharding": "channels, no collective", "unmeasured_on_multi_gpu_hardware": len(set(devices)) < G},
           "roofline": {"bound": "hbm", "kernel": dominant_kernel(launched, cfg["kernel"]), "kernels_launched_in_the_timed_region": launched,
                        "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None, "kernel_ms": round(per_gpu_ms, 5),
                        "kernel_ms_method": "host clock around the timed steps (all shards' streams synchronised on both sides) / steps; per GPU when the shards are on different devices"}}
    print(json.dumps(out), flush=True)
    for v in shards:
        v.close()
    batch.close()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 2000; c5: 16 calls of 646 blocks)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps in front of the timed region (default = --settle)")
    ap.add_argument("--settle", type=int, default=None,
                    help="at least this many untimed steps precede the timed region whatever --warmup says (clock settling; "
                         "default 1500 for c2, 300 for c4, 4 for c5); 0 = honour --warmup to the letter")
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c2")
    ap.add_argument("--channels", type=int, default=None, help="channels per GPU (default: the config's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-robustness", action="store_true")
    ap.add_argument("--dry-run", action="store_true", help="no GPU: gloo, empty timed loop (tests the launch plumbing)")
    ap.add_argument("--single-process", action="store_true", help="one process, N GPUs through a sharded batch (asdr_create_sharded)")
    ap.add_argument("--devices", type=str, default=None, help="--single-process: comma-separated device ordinals, one per shard")
    ap.add_argument("--no-host-path", action="store_true", help="skip the h2d_d2h_inclusive measurement")
    ap.add_argument("--autopin-leg", action="store_true", help=argparse.SUPPRESS)   # (child process of host_path: the asdr_host_autopin(1) leg alone, one JSON object)
    ap.add_argument("--device", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--no-configs", action="store_true", help="skip the `configs` object (C3 / C4 share / C5 share beside C2's headline)")
    ap.add_argument("--caller-stream", action="store_true",
                    help="launch on torch's current stream (strict stream order) instead of ASDR_STREAM_BATCH")
    args = ap.parse_args()
    if args.autopin_leg:
        import numpy as np
        print(json.dumps(host_path(np, args.device, args.channels or 65536, only_autopin=True)))
        return
    cfg = CONFIGS[args.config]
    if args.steps is None:
        args.steps = 16 if args.config == "c5" else 2000
    settle_min = cfg["settle"] if args.settle is None else args.settle
    warm_req = args.warmup
    if args.warmup is None:
        args.warmup = 2 if args.config == "c5" else 20

    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if args.single_process:
        if launched:
            sys.stderr.write("bench.py: --single-process under a distributed launcher makes no sense\n"); sys.exit(2)
        return main_single_process(args, cfg, settle_min, warm_req)
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

    # this rank's shard of the job: a contiguous channel range (audiosdr_amd/sharding.py), no data-path collective
    from audiosdr_amd.sharding import shard_range
    if args.channels is not None:
        n_ch, ch0, total_ch = args.channels, rank * args.channels, world * args.channels
    elif cfg["per_gpu"]:
        n_ch, ch0, total_ch = cfg["channels"], rank * cfg["channels"], world * cfg["channels"]
    else:
        lo, hi = shard_range(cfg["channels"], rank, world)
        n_ch, ch0, total_ch = hi - lo, lo, cfg["channels"]
    T = cfg["blocks"]

    batch = None
    stream = None

    def make_batch():
        import audiosdr_amd as A
        b = A.AudioSDRBatch(n_ch, device=local_rank)
        if args.config == "c2":
            configure_c2(b)
        elif args.config == "c4":
            configure_c4(b, lib=A.load_library(), channel0=ch0)
        else:
            configure_c5(b)
            b.capture_open(T * 4)         # the sink is rewound every 4 calls: the bench keeps 4 calls' worth, not a whole slot
        return b

    def _step(batch, i):
        if args.config == "c5":
            if batch.capture_position + T > batch.capture_capacity:
                batch.capture_rewind()
            batch.capture_update_device(dI.data_ptr(), dQ.data_ptr(), T, None, stream)
        else:
            b = i % N_INPUT_BLOCKS
            batch.update_device(dI[b].data_ptr(), dQ[b].data_ptr(), dOut.data_ptr(), 1, stream)

    if not args.dry_run:
        import audiosdr_amd as A
        caller_stream = torch.cuda.current_stream().cuda_stream
        stream = caller_stream if args.caller_stream else A.STREAM_BATCH
        if args.config == "c2":
            dI, dQ = tiled_input(np, torch, dev, n_ch, N_INPUT_BLOCKS, n_ch // 4, channel0=ch0, fc=6290.0, A=0.25)
        elif args.config == "c4":
            # 3584 distinct input channels tiled over the shard (the input does not depend on the mode; the mode is (global channel) mod 7)
            dI, dQ = tiled_input(np, torch, dev, n_ch, N_INPUT_BLOCKS, 3584, channel0=ch0 % 3584, fc=6890.0 - 300, A=0.3, m=0.4,
                                 f2=7500.0, a2=0.15)
        else:
            dI, dQ = tiled_input(np, torch, dev, n_ch, T, n_ch, channel0=ch0, per_block=False, fc=6890.0, A=0.02, noise=0.05)
        if args.config != "c5":
            dOut = torch.empty((n_ch, BLOCK), dtype=torch.int16, device=dev)
        # Clock settling on a scratch batch (module docstring): the measured batch gets exactly --warmup steps.  The measured batch
        # is built FIRST: creating a batch (350 MB of state for c2) idles the GPU for tens of milliseconds, after which the first ~100
        # launches run 10-25 % slow while the clocks come back (tools/warmup_curve.py) -- the settle launches must run right up to
        # the measured batch's first step.
        torch.cuda.synchronize()   # the inputs are torch's work on torch's stream; the steps run on the batch's own (non-blocking) streams
        batch = make_batch()
        scratch = None
        if settle_min > 0:
            scratch = make_batch()
            for i in range(settle_min):
                _step(scratch, i)

    def step(i):
        if batch is not None:
            _step(batch, i)

    def fence():
        if not args.dry_run:
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        if not args.dry_run:
            torch.cuda.synchronize()

    # untimed steps of the measured batch: exactly what --warmup asks for (the clocks were settled on a scratch batch above)
    untimed = args.warmup
    for i in range(untimed):
        step(i)
    fence()
    # Kernel time: ONE HIP-event pair, recorded by the library on the launch stream, around the K steps of the timed region;
    # kernel_ms = elapsed / K = the average time per back-to-back step, gaps between the kernels included.  (A pair around
    # every launch puts two more packets between consecutive kernels: it slows the job by ~5 % and measures its own overhead;
    # that per-launch figure is taken AFTER the timed region, from 40 further launches, and reported beside it.)
    fence()
    if batch is not None:
        A.binding.kernels_launched(reset=True)   # the launch census of the timed region names the kernel the roofline object is about
        batch.region_timing_begin(stream)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    if batch is not None:
        region_ms, region_calls = batch.region_timing_end()     # records the stop event, synchronises the stream
    fence()
    t1 = time.perf_counter()
    wall = t1 - t0
    launched = A.binding.kernels_launched() if batch is not None else {}
    if not args.dry_run and scratch is not None:
        scratch.close()
    k_ms = region_ms / max(1, region_calls) if batch is not None else 0.0
    pair_ms = None
    ordered = None
    steady = None
    lane_calls = batch.lane_calls() if batch is not None else 0
    if batch is not None and args.config == "c2" and args.steps < 1000:
        # A short timed region (the driver's --warmup 5 --steps 20) is the first few blocks of a FRESH receiver bank: the AGC's envelope
        # is still creeping up to the signal's peaks (attacking samples in most eight-sample chunks; profiles/README.md "The first blocks
        # of a fresh batch").  `value` stays what those steps took; the same batch after 500 further steps is reported beside it.
        ms_s = measure_region(batch, step, stream, 500, 1500)
        ach_s = ALGO_BYTES_PER_BLOCK * n_ch / (ms_s * 1e-3) / 1e9
        steady = {"ms_per_step": round(ms_s, 5), "frac": round(ach_s / HBM_PEAK_GBS, 4), "Msamples_per_s": round(n_ch * BLOCK / ms_s / 1e3, 1),
                  "what": "the same batch, same launch form, after the timed region: 500 untimed steps, then one HIP-event pair around 1,500 "
                          "steps -- the receiver bank in its steady state (the timed region above is blocks %d..%d of a fresh bank)"
                          % (args.warmup, args.warmup + args.steps - 1)}
    if batch is not None and args.config == "c2":
        if not args.caller_stream:   # the same steps in strict stream order, on a caller's stream
            def cs_step(i):
                b = i % N_INPUT_BLOCKS
                batch.update_device(dI[b].data_ptr(), dQ[b].data_ptr(), dOut.data_ptr(), 1, caller_stream)
            ms_o = measure_region(batch, cs_step, caller_stream, 50, min(args.steps, 600))
            ach_o = ALGO_BYTES_PER_BLOCK * n_ch / (ms_o * 1e-3) / 1e9
            ordered = {"kernel_ms": round(ms_o, 5), "frac": round(ach_o / HBM_PEAK_GBS, 4), "Msamples_per_s": round(n_ch * BLOCK / ms_o / 1e3, 1),
                       "what": "the same calls on a caller's stream: every call is ordered behind the whole previous call, one kernel "
                               "per step, no lanes (one HIP-event pair around %d steps after 50 untimed ones)" % min(args.steps, 600)}
        batch.kernel_timing_begin(40)
        for i in range(40):
            batch.update_device(dI[i % N_INPUT_BLOCKS].data_ptr(), dQ[i % N_INPUT_BLOCKS].data_ptr(), dOut.data_ptr(), 1, caller_stream)
        pair_ms = float(np.mean(batch.kernel_timing_end(40)))
    if dist is not None:
        tw = torch.tensor([wall], dtype=torch.float64, device=dev)
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        wall = float(tw.item())
    pipeline_calls = batch.stream_pipeline_launches() if batch is not None else None

    if rank == 0:
        samples = float(total_ch) * T * BLOCK * args.steps
        algo = cfg["algo"] * n_ch * T                      # this rank's algorithmic bytes per step
        ach = algo / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        # HBM bytes per launch: NOT measured in this run -- copied from the committed rocprofv3 PMC passes of this same command
        # (tools/prof_pmc2.sh); dropped when that file was taken from a different build of the library
        traffic, traffic_source = None, None
        try:
            with open(os.path.join(ROOT, PMC_FILE)) as f:
                pj = json.load(f)
            if args.config == "c2" and n_ch == CHANNELS_PER_GPU and not args.dry_run:
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
        # The kernel against its ISSUE roofline (profiles/issue_latest.json, tools/issue_model.py: builder-run counter passes of this command on
        # a caller's stream, stamped with the sources they were taken from): vector-unit cycles per wave by instruction class -> the time
        # the launch would take if no vector unit ever idled.  Dropped when the file is from other sources.
        issue = None
        try:
            with open(os.path.join(ROOT, ISSUE_FILE)) as f:
                ij = json.load(f)
            if args.config == "c2" and n_ch == CHANNELS_PER_GPU and not args.dry_run:
                from audiosdr_amd import build as _build
                if ij.get("source_sha256") == _build.source_sha256():
                    floor = ij.get("valu_floor_ms")
                    issue = {k: ij.get(k) for k in ("kernel", "valu_instructions_per_wave", "valu_instructions_per_wave_by_class", "cycles_per_instruction_by_class",
                                                    "valu_cycles_per_wave", "wave_lifetime_cycles_in_the_full_launch", "wave_cycles_issuing_frac",
                                                    "wave_cycles_issue_stalled_frac", "wave_cycles_parked_at_a_wait_frac", "shader_clock_ghz_during_the_launch",
                                                    "resident_waves_per_simd", "valu_floor_ms")}
                    issue["valu_busy_frac_this_run"] = round(floor / k_ms, 4) if floor and k_ms > 0 else None
                    issue["source"] = "%s (rocprofv3 --pmc passes of this command on a caller's stream + tools/ubench/valu_rate.hip's cycles per instruction class)" % ISSUE_FILE
                    issue["reading"] = ("valu_floor_ms = waves x valu_cycles_per_wave / (1024 SIMDs x clock): the launch with every vector unit busy every cycle. "
                                        "The rest of the measured time is three resident waves per SIMD working through their own dependent streams "
                                        "(wave_cycles_* fractions): occupancy (168 -> 128 VGPRs and 12.4 -> 10 KB of LDS per wave for a fourth wave) is the lever left, "
                                        "not arithmetic -- DESIGN.md 5")
        except Exception:
            pass
        # The kernel against its LATENCY roofline (profiles/roofline_latency_latest.json, tools/latency_model.py: per phase, the dependent instruction
        # stream -- what a lone workgroup needs for it -- x the measured issue interval of a wave that shares its SIMD with two others running
        # the same mix, tools/ubench/issue_rate.hip).  Dropped when the file is from other sources.
        latency = None
        try:
            with open(os.path.join(ROOT, LATENCY_FILE)) as f:
                lj = json.load(f)
            if args.config == "c2" and n_ch == CHANNELS_PER_GPU and not args.dry_run:
                from audiosdr_amd import build as _build
                if lj.get("source_sha256") == _build.source_sha256():
                    latency = {k: lj.get(k) for k in ("kernel", "model", "wave_lifetime_cycles", "model_over_measured", "ms_per_step", "shader_clock_ghz_assumed",
                                                      "waves", "resident_waves")}
                    latency["phases"] = [{k: p.get(k) for k in ("phase", "mix", "instructions", "lone_cycles", "cycles_per_instruction_at_1_2_3_waves_per_simd",
                                                                "three_active_waves_cycles", "full_launch_cycles") if k in p} for p in lj.get("phases", [])]
                    latency["source"] = "%s (tools/latency_model.py on one GPU box: tools/ubench/issue_rate + tools/timeline.py mw as a lone workgroup and inside the full launch; the timeline build carries 28 marks per wave and runs on a caller's stream)" % LATENCY_FILE
                    latency["reading"] = ("floor_lone_stream: every wave gets through its own dependent stream as if alone on its SIMD (nothing shortens that without removing instructions); "
                                          "three_always_active_waves: the same streams at the issue interval measured for three waves sharing a SIMD; the measured lifetime lies between "
                                          "the two, the north-star's 0.0891 ms needs it within 28 % of the lone stream -- DESIGN.md 5")
        except Exception:
            pass
        workload = {
            "c2": "C2: SSB (USB) demod, %d channels/GPU x 1 block/step, NB+IF+mixer+Hilbert+audio IIR+AGC" % n_ch,
            "c4": "C4: mixed modes (channel mod 7: LSB, USB, CW_LSB, CW_USB, AM, SAM, WSPR) + ALS notch + blanker at 10 dB, %d channels "
                  "in all over %d GPU(s) = %d on this one, 1 block/step" % (total_ch, world, n_ch),
            "c5": "C5: WSPR receivers (BareBonesWSPR.ino settings), %d in all over %d GPU(s) = %d on this one, one step = a %d-block "
                  "call (1/64 of a 2-minute slot) into the capture sink" % (total_ch, world, n_ch, T),
        }[args.config]
        inputs = {
            "c2": "tone + LCG noise per channel, no impulses, all channels configured together (one mixer phase: every wave hits the "
                  "local-oscillator cache), steady tone (the AGC hangs between envelope peaks): the best case of four data-dependent "
                  "fast paths -- `robustness` holds the same chain with each of them defeated",
            "c4": "carrier + 40 % AM + a second tone 610 Hz away at -6 dB (the notch has work) + LCG noise",
            "c5": "weak tone (0.02) under noise (0.05); the same resident 646-block period is streamed",
        }[args.config]
        out = {
            "metric": "Msamples/s through full SSB demod chain, batched 128-sample blocks",
            "value": round(samples / wall / 1e6, 2), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": untimed, "ms_per_step": round(wall / args.steps * 1e3, 5), "higher_is_better": True,
            "scaling": cfg["scaling"], "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload, "name": args.config,
                       "warmup_requested": warm_req, "clock_settle_launches_on_a_scratch_batch": 0 if args.dry_run else settle_min,
                       "channels_per_gpu": n_ch, "channels_total": total_ch, "blocks_per_step": T, "block": BLOCK,
                       "sharding": "channels, no collective", "input": inputs},
            "roofline": {"bound": "hbm", "kernel": dominant_kernel(launched, cfg["kernel"]), "kernels_launched_in_the_timed_region": launched,
                         "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                         "kernel_ms": round(k_ms, 5),
                         "kernel_ms_method": "one HIP-event pair on the launch stream around the %d timed steps / %d" % (args.steps, args.steps),
                         "launch_stream": "caller's stream (strict stream order)" if args.caller_stream else
                                          "ASDR_STREAM_BATCH: the batch's own streams; %d of the %d untimed + timed calls ran as two never-joined lanes (halves of the channel range)" % (lane_calls, untimed + args.steps),
                         "caller_stream_ordered": ordered,
                         "steady_state": steady,
                         "kernel_ms_event_pair_per_launch": None if pair_ms is None else round(pair_ms, 5),
                         "algorithmic_bytes_per_channel_block": round(cfg["algo"], 1),
                         "algorithmic_bytes_per_launch": int(round(algo))},
        }
        if issue is not None:
            out["roofline_issue"] = issue
        if latency is not None:
            out["roofline_latency"] = latency
        if args.config == "c2":
            out["roofline"]["hbm_read_share_frac"] = round(ALGO_READ_BYTES_PER_BLOCK * n_ch / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if k_ms > 0 else 0.0
        if args.config == "c5":
            out["roofline"]["regime"] = "latency-bound (64 waves of dependent work per GPU share); the HBM fraction is reported as SURVEY 8d asks"
            out["config"]["calls_run_as_block_pipeline"] = pipeline_calls
            if k_ms > 0:
                out["config"]["times_real_time"] = round(T * BLOCK / 44100.0 / (k_ms * 1e-3), 1)
        if args.dry_run:
            out["dry_run"] = True
        if world == 1 and args.config == "c2" and not args.dry_run and not (args.no_robustness and args.no_host_path and args.no_configs):
            if batch is not None:
                batch.close(); batch = None
            del dI, dQ
            if not args.no_host_path:
                out["h2d_d2h_inclusive"] = host_path(np, local_rank, n_ch)
            if not args.no_robustness:
                out["robustness"] = robustness(np, torch, dev, local_rank, n_ch, stream, 300)
            if not args.no_configs:
                out["configs"] = other_configs(np, torch, dev, local_rank)
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
