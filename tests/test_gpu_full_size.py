"""BASELINE configs 3, 4 and 5 at their FULL per-GPU sizes through the C ABI (the small cases of tests/cases.py cover the same
settings block by block with taps).  Inputs are a few thousand distinct channels tiled over the batch on the device; checks:
sampled channels bit-for-bit against the CPU oracle, the tiled-duplicate property (channel c == channel c mod uniq for the
whole batch: channels never interact and the wave schedule does not matter), and the domain facts the configs are about
(SAM lock fraction, capture-sink contents)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
BLOCK = 128


def _device_tiles(torch, I, Q, n_ch):
    """[uniq][n_blk][128] host arrays -> per-block device tensors [n_ch][128] (tiled)."""
    uniq, n_blk = I.shape[0], I.shape[1]
    reps = (n_ch + uniq - 1) // uniq
    dI = [torch.from_numpy(np.ascontiguousarray(I[:, b])).cuda().repeat(reps, 1)[:n_ch].contiguous() for b in range(n_blk)]
    dQ = [torch.from_numpy(np.ascontiguousarray(Q[:, b])).cuda().repeat(reps, 1)[:n_ch].contiguous() for b in range(n_blk)]
    return dI, dQ


def _assert_tiled(torch, dOut, uniq, what):
    n_ch = dOut.shape[0]
    ref = dOut[:uniq]
    for r0 in range(uniq, n_ch, uniq):
        part = dOut[r0:r0 + uniq]
        assert bool(torch.equal(part, ref[:part.shape[0]])), "%s: channels %d.. differ from their duplicates" % (what, r0)


def test_full_size_c3_batch(gpu, ao):
    """C3: SAM + PLL + AGC, 262,144 channels on one GPU.  Lock fraction 1.0 after 12 blocks, sampled channels
    vs the oracle (all 12 blocks), every channel equal to its duplicate."""
    import torch
    from audiosdr_amd.synth import make_iq
    n_ch, uniq, n_blk = 262144, 3584, 12

    def cfg(s):
        s.setDemodMode(5); s.setNoiseBlankerThresholdDb(10.0); s.enableAudioFilter(); s.setAudioFilter(0)

    fc = 6890.0 + (np.arange(uniq) % 7 - 3) * 50.0
    I, Q = make_iq(uniq, n_blk, fc=fc, A=0.3, m=0.5, fm=400.0)
    dI, dQ = _device_tiles(torch, I, Q, n_ch)
    batch = gpu.AudioSDRBatch(n_ch)
    cfg(batch)
    dOut = torch.empty((n_ch, BLOCK), dtype=torch.int16, device="cuda")
    sample = [0, 3, 1234, 3583]
    got = {c: [] for c in sample}
    for b in range(n_blk):
        batch.update_device(dI[b].data_ptr(), dQ[b].data_ptr(), dOut.data_ptr(), 1, 0)
        batch.synchronize()
        _assert_tiled(torch, dOut, uniq, "C3 block %d" % b)
        for c in sample:
            got[c].append(dOut[c + uniq * 70].cpu().numpy().copy())
    for c in sample:
        o = ao.OracleSDR(pll_wrap_bound=False); cfg(o)   # the DEFAULT oracle: the reference's unbounded wrap loops, which must never stall here
        want = o.update(I[c], Q[c]).reshape(n_blk, BLOCK)
        assert np.array_equal(np.stack(got[c]), want), "C3 channel %d" % c
        assert o.pll_stalled() == 0
    st = batch.read_status()
    assert float(st["sam_locked"].mean()) == 1.0
    batch.close()


def test_c4_share(gpu, ao):
    """C4: one GPU's share (131,072 channels) of the 1M-channel mixed-mode batch: mode = c mod 7, ALS notch on, blanker 10 dB."""
    import torch
    from audiosdr_amd.synth import make_iq
    n_ch, uniq, n_blk = 131072, 3584, 8      # uniq is a multiple of 7: tiling keeps the mode pattern
    I, Q = make_iq(uniq, n_blk, fc=6890.0 - 300, A=0.3, m=0.4, f2=7500.0, a2=0.15)
    dI, dQ = _device_tiles(torch, I, Q, n_ch)
    batch = gpu.AudioSDRBatch(n_ch)
    L = gpu.load_library()
    for c in range(n_ch):
        L.asdr_setDemodMode(batch._h, c, c % 7)
    batch.enableALSfilter(); batch.setNoiseBlankerThresholdDb(10.0)
    dOut = torch.empty((n_ch, BLOCK), dtype=torch.int16, device="cuda")
    sample = [0, 1, 2, 3, 4, 5, 6, 3583]
    got = {c: [] for c in sample}
    for b in range(n_blk):
        batch.update_device(dI[b].data_ptr(), dQ[b].data_ptr(), dOut.data_ptr(), 1, 0)
        batch.synchronize()
        _assert_tiled(torch, dOut, uniq, "C4 block %d" % b)
        for c in sample:
            got[c].append(dOut[c + uniq * 30].cpu().numpy().copy())
    for c in sample:
        o = ao.OracleSDR()
        o.setDemodMode(c % 7); o.enableALSfilter(); o.setNoiseBlankerThresholdDb(10.0)
        want = o.update(I[c], Q[c]).reshape(n_blk, BLOCK)
        assert np.array_equal(np.stack(got[c]), want), "C4 channel %d (mode %d)" % (c, c % 7)
    batch.close()


def test_c5_share(gpu, ao):
    """C5: one GPU's share (512 receivers) with the BareBonesWSPR.ino:87-102,129 settings, 2,048 consecutive blocks streamed
    256 per launch into the capture sink; sampled receivers' whole capture rows vs the oracle."""
    import torch
    from audiosdr_amd.synth import make_iq
    n_ch, T, launches = 512, 256, 8

    def cfg(s):
        s.enableAGC(); s.setAGCmode(2); s.disableALSfilter(); s.disableNoiseBlanker(); s.setNoiseBlankerThresholdDb(10.0)
        s.setInputGain(1.0); s.setOutputGain(0.5); s.setIQgainBalance(1.020); s.setAudioFilter(2); s.setDemodMode(6); s.setMute(0)

    I, Q = make_iq(n_ch, T, fc=6890.0, A=0.02, noise=0.05)
    dI, dQ = torch.from_numpy(I).cuda(), torch.from_numpy(Q).cuda()
    batch = gpu.AudioSDRBatch(n_ch)
    cfg(batch)
    batch.capture_open(T * launches)
    for _ in range(launches):          # the same resident 256-block period is streamed 8 times
        batch.capture_update_device(dI.data_ptr(), dQ.data_ptr(), T)
    batch.synchronize()
    assert batch.capture_position == T * launches
    for c in (0, 17, 255, 511):
        o = ao.OracleSDR(); cfg(o)
        want = o.update(np.tile(I[c], (launches, 1)), np.tile(Q[c], (launches, 1)))
        assert np.array_equal(batch.capture_read(c).reshape(-1), want.reshape(-1)), "C5 receiver %d" % c
    batch.close()


@pytest.mark.parametrize("adverse", ["divergent_mixer_phases", "all_three"])
def test_full_size_c2_divergent_phases(gpu, ao, adverse):
    """BASELINE config 2 at full size on the adverse cases bench.py's `robustness` object times (bench.ROBUSTNESS_CASES): every wave
    holds 8 channels with 8 different mixer phases (channel c switched LSB -> USB after c mod 8 blocks, AudioSDR.cpp:187-222: the
    local-oscillator cache and the wave-uniform mixer both miss, AudioSDR.h:508-526); `all_three` adds an impulse in every block
    (the blanker's general path, AudioSDR.cpp:606-650) and AGC hang time 0 (the AGC's general form in every
    chunk, AudioSDR.cpp:404-436).  Sampled channels -- all 8 lanes' worth of two waves, the last wave -- bit-for-bit against the
    oracle driven through the same setter sequence; the tiled-duplicate property for every channel."""
    import os
    import sys
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import bench
    from audiosdr_amd.synth import make_iq
    case = bench.ROBUSTNESS_CASES[adverse]
    n_ch, uniq, n_blk = 65536, 2048, 14
    sig = dict(fc=6290.0, A=0.25)
    if case["impulses"]:
        sig.update(impulse_every=128)
    I, Q = make_iq(uniq, n_blk, **sig)
    dI, dQ = _device_tiles(torch, I, Q, n_ch)
    batch = gpu.AudioSDRBatch(n_ch)
    dOut = torch.empty((n_ch, BLOCK), dtype=torch.int16, device="cuda")
    sample = list(range(0, 16)) + [777, 2040, 2047]
    orcs = {c: ao.OracleSDR() for c in sample}
    got = {c: [] for c in sample}
    want = {c: [] for c in sample}

    def step(b):
        batch.update_device(dI[b].data_ptr(), dQ[b].data_ptr(), dOut.data_ptr(), 1, 0)
        batch.synchronize()
        _assert_tiled(torch, dOut, uniq, "%s block %d" % (adverse, b))
        for c in sample:
            got[c].append(dOut[c + uniq * 31].cpu().numpy().copy())
            want[c].append(orcs[c].update(I[c, b], Q[c, b]))

    # the stagger of bench.stagger_divergent_phases, with the sampled oracles switched at the same blocks
    batch.setDemodMode(0)
    for o in orcs.values():
        o.setDemodMode(0)
    L = gpu.load_library()
    for j in range(8):
        for c in range(j, n_ch, 8):
            L.asdr_setDemodMode(batch._h, c, 1)
        for c, o in orcs.items():
            if c % 8 == j:
                o.setDemodMode(1)
        step(j)
    for s in [batch] + list(orcs.values()):
        bench.configure_c2(s)
        if case["agc"]:
            s.setAGChangTime(0.0)
    for b in range(8, n_blk):
        step(b)
    for c in sample:
        assert np.array_equal(np.stack(got[c]), np.stack(want[c])), "%s channel %d" % (adverse, c)
    st = batch.read_status()
    if case["impulses"]:
        assert int(st["nb_detected"].sum()) > n_ch // 2
    batch.close()


def _scatter_group(c, n_groups):
    """Settings group of channel c: a multiplicative hash, so that neither the group pattern nor its combination with the
    tiled inputs repeats with any period the wave schedule could line up with."""
    return (((c * 2654435761) & 0xFFFFFFFF) >> 16) % n_groups


_SCATTER_CONFIGS = [
    # (mode, audio filter or None, blanker dB or None, ALS (M, lambda, D, peak, static) or None, AGC hang ms or None)
    (1, 0, 10.0, None, None), (0, 3, None, None, 20.0), (2, None, 10.0, (55, 0.5, 3.0, False, False), None),
    (3, 5, 6.0, None, 0.0), (4, 0, 10.0, None, None), (4, None, None, (55, 0.5, 3.0, True, False), None),
    (5, 0, 10.0, None, None), (5, None, 10.0, (32, 0.5, 1.0, False, False), None), (6, 2, None, None, None),
    (1, None, 3.0, (100, 0.05, 7.0, False, False), None), (1, 7, 10.0, (55, 0.5, 3.0, False, True), 1.0), (0, None, None, None, None),
    (6, 2, 20.0, (64, 0.5, 1.0, False, False), None),
]


def _apply_scatter_config(s, cfg, ch=None):
    mode, af, nb_db, als, hang = cfg
    kw = {} if ch is None else {"ch": ch}
    s.setDemodMode(mode, **kw)
    if af is not None:
        s.enableAudioFilter(**kw); s.setAudioFilter(af, **kw)
    if nb_db is not None:
        s.enableNoiseBlanker(**kw); s.setNoiseBlankerThresholdDb(nb_db, **kw)
    else:
        s.disableNoiseBlanker(**kw)
    if als is not None:
        M, lam, D, peak, static = als
        s.enableALSfilter(**kw); s.setALSfilterParams(M, lam, D, **kw)
        if peak: s.setALSfilterPeak(**kw)
        if static: s.setALSfilterStatic(**kw)
    if hang is not None:
        s.setAGChangTime(hang, **kw)


def test_full_size_scattered_settings(gpu, ao):
    """131,072 channels whose settings group is a HASH of the channel index (13 groups covering every kernel kind: plain, SAM as
    three launches, ALS on compact and long rows, SAM + ALS, static and peak ALS) and whose inputs are 1,021 distinct rows tiled
    over the batch (1,021 is prime: the (group, input) combinations do not repeat with the tiling).  Unlike the tiled configs
    above, which channels share a wave -- and which end up in the remainders' mixed waves next to a different-key neighbour --
    differs all over the batch.  Checks per block: (1) every channel equals the first channel with the same (group, input row),
    which sits in a different wave with different neighbours; (2) 39 sampled channels (3 per group, spread over the batch) bit-for-bit
    against the oracle.  AudioSDR.cpp:39-168 (channels never interact)."""
    import torch
    from audiosdr_amd.synth import make_iq
    n_ch, uniq, n_blk = 131072, 1021, 6
    n_groups = len(_SCATTER_CONFIGS)
    fc = 6890.0 + (np.arange(uniq) % 11 - 5) * 37.0
    I, Q = make_iq(uniq, n_blk, fc=fc, A=0.3, m=0.4, fm=350.0, f2=fc + 640.0, a2=0.1, impulse_every=700)
    dI, dQ = _device_tiles(torch, I, Q, n_ch)
    grp = np.array([_scatter_group(c, n_groups) for c in range(n_ch)], dtype=np.int64)
    batch = gpu.AudioSDRBatch(n_ch)
    for c in range(n_ch):
        _apply_scatter_config(batch, _SCATTER_CONFIGS[grp[c]], ch=c)
    # representative (first) channel of every (group, input row) combination
    combo = grp * uniq + (np.arange(n_ch) % uniq)
    _, first, inverse = np.unique(combo, return_index=True, return_inverse=True)
    rep = torch.from_numpy(first[inverse]).cuda()
    rng = np.random.default_rng(5)
    sample = []
    for g in range(n_groups):
        members = np.nonzero(grp == g)[0]
        sample += [int(members[0]), int(members[len(members) // 2 + int(rng.integers(0, 50))]), int(members[-1])]
    dOut = torch.empty((n_ch, BLOCK), dtype=torch.int16, device="cuda")
    got = {c: [] for c in sample}
    for b in range(n_blk):
        batch.update_device(dI[b].data_ptr(), dQ[b].data_ptr(), dOut.data_ptr(), 1, 0)
        batch.synchronize()
        if b == 0:   # (the layout is the one the call's flush built)
            layout = batch.schedule_layout()
            assert layout["remainders"] > 0, "the schedule has no mixed waves: the test would not cover what it is for"
            assert min(layout[k] for k in ("plain", "sam", "als_long", "als_compact", "sam_als")) > 0, layout
        same = (dOut == dOut[rep]).all(dim=1)
        assert bool(same.all()), "block %d: channel %d differs from channel %d (same settings, same input)" % (
            b, int((~same).nonzero()[0]), int(rep[int((~same).nonzero()[0])]))
        for c in sample:
            got[c].append(dOut[c].cpu().numpy().copy())
    for c in sample:
        o = ao.OracleSDR()
        _apply_scatter_config(o, _SCATTER_CONFIGS[grp[c]])
        want = o.update(I[c % uniq], Q[c % uniq]).reshape(n_blk, BLOCK)
        assert np.array_equal(np.stack(got[c]), want), "channel %d (group %d)" % (c, grp[c])
    batch.close()


def test_c4_whole_job_on_one_gpu(gpu, ao):
    """BASELINE config 4 WHOLE on one GPU (what `bench.py --config c4` times at N = 1): 1,048,576 channels (the ABI's maximum),
    mode = channel mod 7, ALS notch, blanker at 10 dB; 6 blocks: 4 on the batch's own streams (lanes: halves of every sub-range), one
    on a caller's stream, and a final 2-block call.  Every channel equal to its duplicate (3,584 distinct inputs tiled, which keeps
    the mode pattern: 3,584 = 7 x 512), two channels per mode bit-for-bit against the oracle -- from tiles at both ends of the batch,
    i.e. from both lanes and across the 32-bit row-offset range."""
    import torch
    from audiosdr_amd.synth import make_iq
    n_ch, uniq, n_blk = 1048576, 3584, 6
    I, Q = make_iq(uniq, n_blk, fc=6890.0 - 300, A=0.3, m=0.4, f2=7500.0, a2=0.15, impulse_every=3000)
    dI, dQ = _device_tiles(torch, I, Q, n_ch)
    batch = gpu.AudioSDRBatch(n_ch)
    L = gpu.load_library()
    for c in range(n_ch):
        L.asdr_setDemodMode(batch._h, c, c % 7)
    batch.enableALSfilter(); batch.setNoiseBlankerThresholdDb(10.0)
    dOut = torch.empty((n_ch, 2, BLOCK), dtype=torch.int16, device="cuda")
    # Oracle sample (round 5: 28 -> 112 channel-positions): 14 channels = two per mode, from the channel positions 0..13 AND 3570..3583 of a
    # tile (the first and the last waves a tile's channels land in after the schedule's sort by mode), in the first tile, the tile around the
    # lanes' cut (tile 146: channel 523,264 -- the halves of every sub-range meet near there) and the last whole tile: two or more channels
    # per (mode, lane, tile-end) combination.  All against the DEFAULT oracle (the reference's unbounded PLL wrap), which must not stall.
    sample = list(range(14)) + list(range(uniq - 14, uniq))
    tiles = [0, 146, 291]                             # tile 292 (from channel 1,046,528) is the batch's last, partial one: 2,048 channels
    assert max(sample) + uniq * max(tiles) < n_ch
    got = {(c, t): [] for c in sample for t in tiles}
    last = {c: [] for c in range(14)}

    def collect(nb):
        batch.synchronize()
        for k in range(nb):
            _assert_tiled(torch, dOut[:, k], uniq, "C4 whole job")
            for (c, t) in got:
                got[(c, t)].append(dOut[c + uniq * t, k].cpu().numpy().copy())
            for c in range(14):                       # ... and the first channels of the partial last tile (the very end of the 32-bit row offsets)
                last[c].append(dOut[c + uniq * 292, k].cpu().numpy().copy())

    for b in range(4):
        batch.update_device_strided(dI[b].data_ptr(), dQ[b].data_ptr(), dOut.data_ptr(), 1, 1, 2, gpu.STREAM_BATCH if b != 2 else 0)
        collect(1)
    assert batch.lane_calls() == 2                    # (the first call flushes the setters; block 2 went to a caller's stream)
    # the last two blocks as ONE two-block call (one launch per block inside it)
    dI2 = torch.stack([dI[4], dI[5]], dim=1).contiguous(); dQ2 = torch.stack([dQ[4], dQ[5]], dim=1).contiguous()
    batch.update_device(dI2.data_ptr(), dQ2.data_ptr(), dOut.data_ptr(), 2, 0)
    collect(2)
    for c in sample:
        o = ao.OracleSDR(pll_wrap_bound=False)
        o.setDemodMode(c % 7); o.enableALSfilter(); o.setNoiseBlankerThresholdDb(10.0)
        want = o.update(I[c], Q[c]).reshape(n_blk, BLOCK)
        assert o.pll_stalled() == 0
        for t in tiles:
            assert np.array_equal(np.stack(got[(c, t)]), want), "C4 channel %d of tile %d (mode %d)" % (c, t, c % 7)
        if c < 14:
            assert np.array_equal(np.stack(last[c]), want), "C4 channel %d of the last tile (mode %d)" % (c, c % 7)
    batch.close()


def test_c5_full_two_minute_slot(gpu, ao):
    """BASELINE config 5, one GPU's share, the WHOLE 2-minute WSPR slot: 512 receivers x 41,344 blocks (BareBonesWSPR.ino:87-102,129
    settings) streamed as 64 calls of 646 blocks into the capture sink (block pipeline), from a resident 646-block period whose
    receivers differ in carrier offset and noise seed.  Four receivers' whole 5.3-million-sample capture rows bit-for-bit against the
    oracle; the other rows through their checksums' distinctness (no two receivers carry the same audio) and the sink's bookkeeping."""
    import torch
    from audiosdr_amd.synth import make_iq
    n_ch, T, calls = 512, 646, 64

    def cfg(s):
        s.enableAGC(); s.setAGCmode(2); s.disableALSfilter(); s.disableNoiseBlanker(); s.setNoiseBlankerThresholdDb(10.0)
        s.setInputGain(1.0); s.setOutputGain(0.5); s.setIQgainBalance(1.020); s.setAudioFilter(2); s.setDemodMode(6); s.setMute(0)

    fc = 6890.0 - 5390.0 + 1500.0 + 1.4648 * (np.arange(n_ch) % 4)      # 4-FSK-like tone offsets
    I, Q = make_iq(n_ch, T, fc=fc, A=0.02, noise=0.05)
    dI, dQ = torch.from_numpy(I).cuda(), torch.from_numpy(Q).cuda()
    batch = gpu.AudioSDRBatch(n_ch)
    cfg(batch)
    batch.capture_open(T * calls)
    assert T * calls == 41344
    for _ in range(calls):
        batch.capture_update_device(dI.data_ptr(), dQ.data_ptr(), T, None, gpu.STREAM_BATCH)
    batch.synchronize()
    assert batch.capture_position == 41344 and batch.stream_pipeline_launches() == calls and batch.stream_pipeline_recoveries() == 0
    for c in (0, 201, 511, 338):
        o = ao.OracleSDR(); cfg(o)
        want = o.update(np.tile(I[c], (calls, 1)), np.tile(Q[c], (calls, 1)))
        assert np.array_equal(batch.capture_read(c).reshape(-1), want.reshape(-1)), "C5 receiver %d" % c
    # every receiver's row is its own (different noise seeds): distinct checksums of the slot's last 64 blocks, every 37th receiver
    sums = set()
    for c in range(0, n_ch, 37):
        row = batch.capture_read(c, first_block=41344 - 64, n_blocks=64)
        sums.add(int(row.astype(np.int64).sum()) * 1000003 + int((row.astype(np.int64) ** 2).sum()))
    assert len(sums) == len(range(0, n_ch, 37))
    batch.close()
