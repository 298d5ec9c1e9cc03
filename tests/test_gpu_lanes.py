"""Lanes (include/asdr.h, ASDR_STREAM_BATCH): a batch runs the two halves of every sub-range of its schedule on two streams of its own that never
wait for each other -- from call to call on ASDR_STREAM_BATCH, inside a multi-block call on a caller's stream.  Same kernels, same
channels, only the launch geometry differs: every output and every status word must equal the ordinary path's, whatever the
sequence of lane calls, ordinary calls, setters (which take a batch off the lanes for one call, or for good when the schedule stops
being one group), stream switches and host-side reads in between."""
import numpy as np
import pytest

from tests.helpers import Hip

pytestmark = pytest.mark.gpu


def _tile(a, n):
    return np.ascontiguousarray(np.tile(a, ((n + a.shape[0] - 1) // a.shape[0], 1, 1))[:n])


def _twin(gpu, n, mode, audio=True):
    out = []
    for lanes in (True, False):
        b = gpu.AudioSDRBatch(n)
        b.setDemodMode(mode)
        if audio:
            b.enableAudioFilter()
        b.setNoiseBlankerThresholdDb(10.0)
        b.set_lanes(lanes)
        out.append(b)
    return out


@pytest.mark.parametrize("mode,n", [(1, 16384 + 8), (4, 9000), (5, 8192)])
def test_lane_calls_equal_ordinary_calls(gpu, ao, mode, n):
    from audiosdr_amd.synth import make_iq
    uniq, total = 96, 21
    fc = 6890.0 - (600.0 if mode == 1 else 0.0) + 20.0 * (np.arange(uniq) % 5)
    bI, bQ = make_iq(uniq, total, fc=fc, A=0.3, m=0.4 if mode != 1 else 0.0, noise=0.02, impulse_every=1700)
    I, Q = _tile(bI, n), _tile(bQ, n)
    hip = Hip()
    s1, s2 = hip.stream(), hip.stream()
    dI, dQ = hip.upload(I), hip.upload(Q)
    A, B = _twin(gpu, n, mode)
    dA, dB = hip.malloc(n * total * 256), hip.malloc(n * total * 256)
    BATCH = gpu.STREAM_BATCH
    pos = 0

    def both(T, stream_a, stream_b=0):
        nonlocal pos
        off = pos * 256
        A.update_device_strided(dI + off, dQ + off, dA + off, T, total, total, stream_a)
        B.update_device_strided(dI + off, dQ + off, dB + off, T, total, total, stream_b)
        pos += T

    lane_calls = 0
    both(1, BATCH)                     # the first call applies the construction's setters and resets: ordinary
    both(1, BATCH); lane_calls += 1
    both(1, BATCH); lane_calls += 1
    both(1, s1, s1)                    # a caller's stream in between: ordered behind the lanes
    both(3, BATCH); lane_calls += 1    # multi-block on the lanes
    both(1, BATCH); lane_calls += 1
    assert A.lane_calls() == lane_calls and B.lane_calls() == 0
    for b in (A, B):
        b.setOutputGain(0.7, ch=5)     # a parameter row changes: one ordinary call (it flushes), then the lanes again
    both(1, BATCH)
    both(2, BATCH); lane_calls += 1
    both(4, s2, s2)                    # a multi-block call on a caller's stream: lanes inside it when it is large enough for one launch per block
    if n >= 8192 or mode == 5:
        lane_calls += 1
    assert A.lane_calls() == lane_calls, (A.lane_calls(), lane_calls)
    st = A.read_status()               # host-side read: synchronises the lanes
    for b in (A, B):
        b.setDemodMode(0, ch=n // 2)   # several settings groups from now on (a general-kernel remainder wave among them): still lanes
    both(2, BATCH)
    both(1, BATCH); lane_calls += 1
    assert A.lane_calls() == lane_calls
    for b in (A, B):
        b.setDemodMode(mode, ch=n // 2)
    both(1, BATCH)
    both(2, BATCH); lane_calls += 1
    assert A.lane_calls() == lane_calls and pos == total
    A.synchronize(); B.synchronize()
    wA = hip.download(dA, (n, total, 128), np.int16)
    wB = hip.download(dB, (n, total, 128), np.int16)
    assert np.array_equal(wA, wB), "%d samples differ" % int((wA != wB).sum())
    sA, sB = A.read_status(), B.read_status()
    for k in sA:
        assert sA[k].tobytes() == sB[k].tobytes(), k
    # ... and the ordinary path is the oracle's: three channels of the first tile, all blocks up to the LSB excursion
    for c in (0, 5, 37):
        o = ao.OracleSDR()
        o.setDemodMode(mode); o.enableAudioFilter(); o.setNoiseBlankerThresholdDb(10.0)
        w1 = o.update(bI[c, :8], bQ[c, :8]).reshape(8, 128)
        if c == 5:
            o.setOutputGain(0.7)
        w2 = o.update(bI[c, 8:], bQ[c, 8:]).reshape(total - 8, 128)
        assert np.array_equal(wA[c], np.concatenate([w1, w2])), c
    hip.free_all(); A.close(); B.close()


def test_ordering_against_a_callers_stream(gpu):
    """asdr_order_after / asdr_order_before: input uploaded asynchronously on the caller's stream, result downloaded on it, the
    batch's calls on ASDR_STREAM_BATCH in between -- repeated with fresh data each round into the SAME device buffers, so that a
    missing ordering edge shows as a stale or torn block."""
    from audiosdr_amd.synth import make_iq
    n, rounds = 16384, 6
    bI, bQ = make_iq(64, rounds, fc=6290.0, A=0.25, noise=0.02, impulse_every=900)
    hip = Hip()
    s = hip.stream()
    A, B = _twin(gpu, n, 1)
    hI, hQ, hO = (gpu.host_alloc((n, 1, 128)) for _ in range(3))
    dI, dQ, dO = hip.malloc(n * 256), hip.malloc(n * 256), hip.malloc(n * 256)
    got = []
    for r in range(rounds):
        hip.sync(s)                                           # (the host buffers are reused: the previous round's copies are done)
        hI[:] = _tile(bI[:, r:r + 1], n); hQ[:] = _tile(bQ[:, r:r + 1], n)
        hip.copy_async(dI, hI.ctypes.data, n * 256, 1, s); hip.copy_async(dQ, hQ.ctypes.data, n * 256, 1, s)
        A.order_after(s)
        A.update_device(dI, dQ, dO, 1, gpu.STREAM_BATCH)
        A.order_before(s)
        hip.copy_async(hO.ctypes.data, dO, n * 256, 2, s)
        hip.sync(s)
        got.append(hO.copy())
    assert A.lane_calls() == rounds - 1
    want = B.update(_tile(bI, n), _tile(bQ, n))
    for r in range(rounds):
        assert np.array_equal(got[r][:, 0], want[:, r]), r
    for a in (hI, hQ, hO):
        gpu.host_free(a)
    hip.free_all(); A.close(); B.close()


def test_lanes_on_a_mixed_schedule(gpu, ao):
    """BASELINE config 4's mix (mode = channel mod 7, ALS notch, blanker at 10 dB): five kernel kinds' sub-ranges and the remainders,
    each cut in two halves; lane 1 has its own local-oscillator cache writers.  Against the ordinary path on every channel and
    against the oracle on two channels per mode."""
    from audiosdr_amd.synth import make_iq
    n, uniq, total = 12000 + 13, 7 * 16, 14
    fc = 6890.0 - 300.0 + 15.0 * (np.arange(uniq) % 4)
    bI, bQ = make_iq(uniq, total, fc=fc, A=0.3, m=0.4, noise=0.02, f2=fc + 610.0, a2=0.15, impulse_every=2100)
    I, Q = _tile(bI, n), _tile(bQ, n)
    hip = Hip()
    dI, dQ = hip.upload(I), hip.upload(Q)
    pair = []
    for lanes in (True, False):
        b = gpu.AudioSDRBatch(n)
        for m in range(7):
            for c in range(m, n, 7):
                b.setDemodMode(m, ch=c)
        b.enableALSfilter(); b.setNoiseBlankerThresholdDb(10.0)
        b.set_lanes(lanes)
        pair.append(b)
    A, B = pair
    dA, dB = hip.malloc(n * total * 256), hip.malloc(n * total * 256)
    pos = 0
    for T, sa in ((1, gpu.STREAM_BATCH), (1, gpu.STREAM_BATCH), (3, gpu.STREAM_BATCH), (1, 0), (2, gpu.STREAM_BATCH), (4, 0), (1, gpu.STREAM_BATCH), (1, gpu.STREAM_BATCH)):
        off = pos * 256
        A.update_device_strided(dI + off, dQ + off, dA + off, T, total, total, sa)
        B.update_device_strided(dI + off, dQ + off, dB + off, T, total, total, 0)
        pos += T
    assert pos == total and A.lane_calls() == 6 and B.lane_calls() == 0      # (the first call flushes; the 1-block call on stream 0 is ordinary)
    A.synchronize(); B.synchronize()
    wA, wB = hip.download(dA, (n, total, 128), np.int16), hip.download(dB, (n, total, 128), np.int16)
    assert np.array_equal(wA, wB), "%d samples differ" % int((wA != wB).sum())
    sA, sB = A.read_status(), B.read_status()
    for k in sA:
        assert sA[k].tobytes() == sB[k].tobytes(), k
    for c in list(range(14)):                       # uniq is a multiple of 7: channel c of the tile has mode c mod 7
        o = ao.OracleSDR()
        o.setDemodMode(c % 7); o.enableALSfilter(); o.setNoiseBlankerThresholdDb(10.0)
        assert np.array_equal(wA[c], o.update(bI[c], bQ[c]).reshape(total, 128)), c
    hip.free_all(); A.close(); B.close()


def test_host_rows_call_behind_running_lanes(gpu):
    """asdr_update (host rows) right behind a long multi-block call that is still running on the lanes, with a setter in between:
    the host path rewrites parameter rows (its flush) and must not start before EVERY lane is done -- found by the launch-form fuzz
    (seed 144: the setter's channel, in the third lane, ran the rest of the running call with the filter the setter had just
    enabled).  16,384 SAM channels x 40 blocks keep the lanes busy for a few milliseconds; the setters (output gain: visible in every
    sample from the block in which the row changes) hit channels of every lane.  (A functional check of the sequence: without the join the
    outcome depends on which lane stream lags -- the fuzz seed failed one run in three, this sequence did not fail in six.)"""
    from audiosdr_amd.synth import make_iq
    n, uniq, total = 16384, 64, 42
    fc = 6890.0 + 25.0 * (np.arange(uniq) % 5)
    bI, bQ = make_iq(uniq, total, fc=fc, A=0.3, m=0.4, noise=0.01)
    I, Q = _tile(bI, n), _tile(bQ, n)
    hip = Hip()
    dI, dQ = hip.upload(I), hip.upload(Q)
    A, B = _twin(gpu, n, 5)
    A.set_lanes(4)      # four lanes on the pool's three streams: one stream lags
    dA, dB = hip.malloc(n * total * 256), hip.malloc(n * total * 256)
    last = (np.ascontiguousarray(I[:, 41:42]), np.ascontiguousarray(Q[:, 41:42]))
    out = []
    for b, d, s in ((A, dA, gpu.STREAM_BATCH), (B, dB, 0)):
        b.update_device_strided(dI, dQ, d, 1, total, total, s)                     # block 0 (flushes the settings)
        b.update_device_strided(dI + 256, dQ + 256, d + 256, 40, total, total, s)  # blocks 1..40: A on its lanes, not waited for
        for c in (5, n // 4 + 3, n // 2 + 11, 3 * n // 4 + 7, n - 1):
            b.setOutputGain(0.25, ch=c)     # (a parameter-row change only: the flush is a few microseconds of host work, the lanes still run)
        out.append(b.update(*last))         # host rows, block 41 -- at once
    outA, outB = out
    assert A.lane_calls() == 1
    assert np.array_equal(outA, outB), "block 41: %d samples differ" % int((outA != outB).sum())
    A.synchronize(); B.synchronize()
    wA, wB = hip.download(dA, (n, total, 128), np.int16)[:, :41], hip.download(dB, (n, total, 128), np.int16)[:, :41]
    bad = np.argwhere(wA != wB)
    assert bad.size == 0, "%d samples differ, first at %s" % (len(bad), bad[0].tolist())
    hip.free_all(); A.close(); B.close()


def test_lanes_with_foreign_streams_alive_and_the_overlap_probe(gpu):
    """Round 5: the lanes rest on the pool's streams running concurrently, which every other stream of the process can change (include/asdr.h,
    "WHAT THE LANES REST ON").  With three torch streams alive and busy beside the batch: (1) the probe (run by the first lane-sized call) has an answer for the device
    (1 concurrent / 0 serialised; -1 only if switched off) and the batch's default follows it; (2) whatever the probe said, lane calls
    and ordinary calls give identical audio (the geometry is a speed matter, never a correctness one)."""
    import os
    import torch
    from audiosdr_amd.synth import make_iq
    n, total = 16384, 6
    foreign = [torch.cuda.Stream() for _ in range(3)]
    junk = [torch.zeros(1 << 20, device="cuda") for _ in foreign]
    for s, t in zip(foreign, junk):
        with torch.cuda.stream(s):
            for _ in range(20):
                t.add_(1.0)
    bI, bQ = make_iq(64, total, fc=6290.0, A=0.25)
    I, Q = _tile(bI, n), _tile(bQ, n)
    A, B = _twin(gpu, n, 1)
    hip = Hip()
    dI, dQ = hip.upload(I), hip.upload(Q)
    dA, dB = hip.malloc(n * total * 256), hip.malloc(n * total * 256)
    for b in range(total):
        for s, t in zip(foreign, junk):                 # the foreign streams stay busy between the calls
            with torch.cuda.stream(s):
                t.mul_(1.0001)
        off = b * 256
        A.update_device_strided(dI + off, dQ + off, dA + off, 1, total, total, gpu.STREAM_BATCH)
        B.update_device_strided(dI + off, dQ + off, dB + off, 1, total, total, 0)
    A.synchronize(); B.synchronize()
    assert A.lane_calls() == total - 1                  # (set_lanes(True) in _twin: the lanes run whatever the probe says)
    wa, wb = hip.download(dA, (n, total, 128), np.int16), hip.download(dB, (n, total, 128), np.int16)
    assert np.array_equal(wa, wb)
    C = gpu.AudioSDRBatch(n)                            # a batch with the DEFAULT setting: its first lane-sized call probes the pool and follows it
    C.setDemodMode(1); C.enableAudioFilter(); C.setNoiseBlankerThresholdDb(10.0)
    dC = hip.malloc(n * total * 256)
    for b in range(3):
        C.update_device_strided(dI + b * 256, dQ + b * 256, dC + b * 256, 1, total, total, gpu.STREAM_BATCH)
    C.synchronize()
    probe = C.lanes_overlap_probe()
    assert probe in ((0, 1) if not os.environ.get("ASDR_NO_LANES_PROBE") else (-1, 0, 1))
    if not os.environ.get("ASDR_NO_LANES"):
        assert C.lanes_enabled() == (probe != 0)
        assert (C.lane_calls() > 0) == (probe != 0)
    wc = hip.download(dC, (n, total, 128), np.int16)
    assert np.array_equal(wc[:, :3], wa[:, :3])
    for s in foreign:
        s.synchronize()
    hip.free_all(); A.close(); B.close(); C.close()
