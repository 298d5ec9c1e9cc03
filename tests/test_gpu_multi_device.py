"""Multi-GPU tests that SWITCH THEMSELVES ON (channels shard embarrassingly over the GPUs of one node, no collective).

Every test here runs on G = min(visible HIP devices, 8) DISTINCT device ordinals and is skipped when the box has one device (every
gpurun box of rounds 1-5 did: the sharded batch had only ever run as several shards on device 0, tests/test_sharded_batch.py).  What
they exercise that one device cannot: per-device `__constant__` table uploads, per-device stream pools and lane probes, `hipSetDevice`
in the persistent shard worker threads, device-local pointers per shard, and the caller's current device being put back.
The one-device test at the bottom runs everywhere: it checks the same properties with all shards on device 0.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def _n_devices():
    try:
        import torch
        return int(torch.cuda.device_count())     # (counting devices does not initialise the GPU on this image)
    except Exception:
        return 0


G = min(_n_devices(), 8)
many = pytest.mark.skipif(G < 2, reason="needs >= 2 HIP devices (this box has %d): the test enables itself on a multi-GPU node" % G)


def _hip_set_device(dev):
    import ctypes as C
    h = C.CDLL("libamdhip64.so")
    assert h.hipSetDevice(int(dev)) == 0


def _hip_get_device():
    import ctypes as C
    h = C.CDLL("libamdhip64.so")
    d = C.c_int(-1)
    assert h.hipGetDevice(C.byref(d)) == 0
    return d.value


def _configure_mix(batch, n):
    for m in range(7):
        for c in range(m, n, 7):
            batch.setDemodMode(m, ch=c)
    batch.enableALSfilter()
    batch.setNoiseBlankerThresholdDb(10.0)


def _input(n, nb, seed=0):
    from audiosdr_amd.synth import make_iq
    fc = 6890.0 - 600.0 + 10.0 * (np.arange(n) % 5)
    return make_iq(n, nb, fc=fc, A=0.25, m=0.3, noise=0.02, impulse_every=1900, f2=fc + 1000.0, a2=0.125, seed0=12345 + seed)


def _host_rows_equal_the_single_batch(gpu, devices):
    n, nb = 1500 + 11 * len(devices) + 5, 4
    I, Q = _input(n, 2 * nb)
    one = gpu.AudioSDRBatch(n, device=devices[0])
    sh = gpu.AudioSDRBatch(n, devices=list(devices))
    _configure_mix(one, n); _configure_mix(sh, n)
    one.enable_taps(); sh.enable_taps()
    _hip_set_device(devices[-1])                       # the caller's current device: must survive the sharded call
    a1 = one.update(I[:, :nb], Q[:, :nb])
    _hip_set_device(devices[-1])
    a2 = sh.update(I[:, :nb], Q[:, :nb])               # host rows: scatter / gather, persistent worker thread per shard
    assert _hip_get_device() == devices[-1], "asdr_update on a sharded batch left the caller on another device"
    assert np.array_equal(a1, a2)
    for c in (0, n // len(devices) - 1, n // len(devices), n - 1, 511):
        for b in (one, sh):
            b.setOutputGain(0.8, ch=c); b.setDemodMode((c + 3) % 7, ch=c); b.setAGChangTime(0.0, ch=c)
    for rep in range(3):                               # the same worker threads serve every call
        a1 = one.update(I[:, nb:], Q[:, nb:]) if rep == 0 else a1
        a2 = sh.update(I[:, nb:], Q[:, nb:]) if rep == 0 else a2
    assert np.array_equal(a1, a2)
    s1, s2 = one.read_status(), sh.read_status()
    for k in s1:
        assert s1[k].tobytes() == s2[k].tobytes(), k
    t1, t2 = one.read_taps(), sh.read_taps()
    for k in t1:
        assert t1[k].tobytes() == t2[k].tobytes(), k
    for g in range(len(devices)):
        assert gpu.load_library().asdr_shard_device(sh._h, g) == devices[g]
    one.close(); sh.close()


@many
@pytest.mark.gpu
def test_sharded_batch_over_distinct_devices_equals_the_single_batch(gpu):
    _host_rows_equal_the_single_batch(gpu, list(range(G)))


@many
@pytest.mark.gpu
def test_shard_handles_with_device_local_pointers_lanes_and_capture(gpu):
    """One host thread drives every GPU through its shard handle with rows resident on THAT GPU, on ASDR_STREAM_BATCH (the lanes of every
    shard); the audio equals a single batch's; the capture sink comes back by global channel."""
    import torch
    devices = list(range(G))
    n_per, nb = 8192 + 24, 3                           # >= 1,024 waves per shard: the lanes' default threshold
    n = n_per * G
    from audiosdr_amd.synth import make_iq
    I, Q = make_iq(2048, nb, fc=6290.0, A=0.25)
    reps = (n + 2047) // 2048
    I = np.tile(I, (reps, 1, 1))[:n]; Q = np.tile(Q, (reps, 1, 1))[:n]
    sh = gpu.AudioSDRBatch(n, devices=devices)
    sh.setDemodMode(1); sh.enableAudioFilter()
    outs = []
    for g in range(G):
        lo, hi = sh.shard_range(g)
        dev = torch.device("cuda", g)
        dI = torch.from_numpy(np.ascontiguousarray(I[lo:hi])).to(dev); dQ = torch.from_numpy(np.ascontiguousarray(Q[lo:hi])).to(dev)
        dO = torch.empty((hi - lo, nb, 128), dtype=torch.int16, device=dev)
        torch.cuda.synchronize(dev)
        v = sh.shard(g)
        for b in range(nb):                            # one block per call, on the shard's own streams
            v.update_device_strided(dI.data_ptr() + b * 256, dQ.data_ptr() + b * 256, dO.data_ptr() + b * 256, 1, nb, nb, gpu.STREAM_BATCH)
        outs.append((v, dO, lo, hi))
    got = np.empty((n, nb, 128), np.int16)
    for v, dO, lo, hi in outs:
        v.synchronize()
        assert v.lanes_overlap_probe() in (0, 1), "the lanes probe did not run on this shard's device"
        if v.lanes_enabled():
            assert v.lane_calls() >= nb - 1, "shard on device %d never used its lanes" % v_dev(v, gpu)
        got[lo:hi] = dO.cpu().numpy()
    one = gpu.AudioSDRBatch(2048, device=0)
    one.setDemodMode(1); one.enableAudioFilter()
    want = one.update(I[:2048], Q[:2048])
    for r0 in range(0, n, 2048):
        part = got[r0:r0 + 2048]
        assert np.array_equal(part, want[:part.shape[0]]), "channels %d.." % r0
    one.close(); sh.close()


def v_dev(v, gpu):
    return gpu.load_library().asdr_shard_device(v._h, 0)


@many
@pytest.mark.gpu
def test_bench_single_process_over_all_devices():
    """bench.py --single-process --gpus G: one process, a sharded batch over G distinct devices; the line says it was measured on
    multi-GPU hardware (`unmeasured_on_multi_gpu_hardware` false)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--single-process", "--gpus", str(G), "--channels", "16384",
                          "--steps", "50", "--warmup", "10", "--settle", "50"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == G and d["config"]["devices"] == list(range(G))
    assert d["config"]["unmeasured_on_multi_gpu_hardware"] is False
    assert d["value"] > 0


@pytest.mark.gpu
def test_sharded_host_rows_on_one_device_with_persistent_workers(gpu):
    """The same host-row check with every shard on device 0 (runs on any box): persistent worker threads, the caller's device kept."""
    _host_rows_equal_the_single_batch(gpu, [0, 0, 0])
