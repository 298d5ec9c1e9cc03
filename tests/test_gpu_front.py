"""GPU parity of the blocks around the hot path through the C ABI of include/asdr_front.h,
bit-for-bit against the CPU oracle (oracle/asdr_front_oracle.c): int16 streams, counters, and the detector's
float32 measurements."""
import numpy as np
import pytest

from helpers import Hip, f32_bits
from test_front_oracle import tone_iq

pytestmark = pytest.mark.gpu
FS = 44100.0


def _streams(n_ch, n_blk, kinds, seed=0):
    """Per-channel I/Q streams [n_ch][n_blk][128]: kind 0 clean tone, 1 Q late, 2 I late, 3 noise only, 4 silence."""
    I = np.zeros((n_ch, n_blk * 128), dtype=np.int16); Q = np.zeros_like(I)
    rng = np.random.default_rng(seed)
    for c in range(n_ch):
        k = kinds[c % len(kinds)]
        f = 6000.0 + 431.0 * c      # a one-sample skew only fails the 10x image-ratio test above ~4.4 kHz
        if k in (0, 1, 2):
            I[c], Q[c] = tone_iq(n_blk, f, amp=0.2 + 0.01 * (c % 5), q_delay=(0, 1, -1)[k], noise=0.002, seed=seed + c)
        elif k == 3:
            I[c] = (rng.standard_normal(n_blk * 128) * 1500).astype(np.int16)
            Q[c] = (rng.standard_normal(n_blk * 128) * 1500).astype(np.int16)
    return I.reshape(n_ch, n_blk, 128), Q.reshape(n_ch, n_blk, 128)


def _cmp_state(batch, orcs, floats=True):
    st = batch.read_state()
    for c, o in enumerate(orcs):
        s = o.state()
        for k in ("correction", "saved_sample", "failure_count", "success_count", "auto_detect", "swap"):
            assert int(st[k][c]) == s[k], (k, c, int(st[k][c]), s[k])
        if floats and s["auto_detect"]:
            assert int(st["max_line"][c]) == s["max_line"] and int(st["strong"][c]) == s["strong"], c
            for k in ("max_power", "avg_power", "ratio"):
                a, b = np.float32(st[k][c]), np.float32(s[k])
                assert f32_bits(a) == f32_bits(b) or (np.isnan(a) and np.isnan(b)), (k, c, a, b)


@pytest.mark.parametrize("n_ch", [1, 3, 4, 5, 67])
def test_pre_fixed_corrections_and_swap(gpu, ao, n_ch):
    """Corrections -1/0/+1/2 x swap on/off per channel, 5 blocks in one call then 3 calls of 1: int16 bit-exact."""
    I, Q = _streams(n_ch, 8, [0, 3], seed=11)
    b = gpu.AudioSDRpreProcessorBatch(n_ch)
    orcs = [ao.OraclePreProcessor() for _ in range(n_ch)]
    for c in range(n_ch):
        corr, sw = (-1, 0, 1, 2)[c % 4], (c // 4) % 2
        b.setI2SerrorCompensation(corr, ch=c); orcs[c].setI2SerrorCompensation(corr)
        b.swapIQ(sw, ch=c); orcs[c].swapIQ(sw)
    want = [o.update(I[c], Q[c]) for c, o in enumerate(orcs)]
    gi, gq = b.update(I[:, :5], Q[:, :5])
    parts_i, parts_q = [gi], [gq]
    for k in range(5, 8):
        gi, gq = b.update(I[:, k:k + 1], Q[:, k:k + 1])
        parts_i.append(gi); parts_q.append(gq)
    gi, gq = np.concatenate(parts_i, axis=1), np.concatenate(parts_q, axis=1)
    for c in range(n_ch):
        assert np.array_equal(gi[c].reshape(-1), want[c][0]), "I ch %d" % c
        assert np.array_equal(gq[c].reshape(-1), want[c][1]), "Q ch %d" % c
    _cmp_state(b, orcs)
    b.close()


def test_pre_autodetect_block_by_block(gpu, ao):
    """Detector on, 36 blocks one call per block: streams, counters and the float32 measurements after every block."""
    n_ch, n_blk = 10, 36
    I, Q = _streams(n_ch, n_blk, [0, 1, 2, 3, 4], seed=3)
    b = gpu.AudioSDRpreProcessorBatch(n_ch)
    orcs = [ao.OraclePreProcessor() for _ in range(n_ch)]
    b.startAutoI2SerrorDetection()
    for o in orcs:
        o.startAutoI2SerrorDetection()
    for k in range(n_blk):
        gi, gq = b.update(I[:, k:k + 1], Q[:, k:k + 1])
        for c, o in enumerate(orcs):
            wi, wq = o.update(I[c, k], Q[c, k])
            assert np.array_equal(gi[c, 0], wi) and np.array_equal(gq[c, 0], wq), (k, c)
        _cmp_state(b, orcs)
    corr = [o.getI2SerrorCompensation() for o in orcs]
    assert corr[0] == 0 and corr[1] == 1 and corr[2] == -1      # the three skews were told apart
    b.close()


def test_pre_detector_ties_and_degenerate_spectra(gpu, ao):
    """The first-maximum scan runs in parallel on the GPU (lanes scan eight lines each, then exchange): the cases where ORDER decides
    must come out as in the reference's sequential scan (AudioSDRpreProcessor.cpp:98-104) -- silence (no line beats the initial 0.0:
    line 0, and the ratio divides by buffer[128]), an impulse at sample 0 (every line the same power, exactly: the first of lines
    5..122 wins), an impulse elsewhere (flat magnitudes with rounding scatter), real-only tones (lines k and 128 - k mirror each
    other), a full-scale DC block (the strongest lines are outside 5..122), maximum in the first / last scanned line."""
    n_blk = 3
    t = np.arange(128)
    blocks = []
    z = np.zeros(128, np.int16)
    imp0 = z.copy(); imp0[0] = 20000
    imp9 = z.copy(); imp9[9] = -12345
    blocks += [(z, z), (imp0, z), (z, imp0), (imp0, imp0), (imp9, z), (imp9, imp9)]
    for k in (5, 6, 64, 121, 122, 123, 4, 20):
        c = np.round(9000 * np.cos(2 * np.pi * k * t / 128)).astype(np.int16)
        sn = np.round(9000 * np.sin(2 * np.pi * k * t / 128)).astype(np.int16)
        blocks += [(c, z), (c, sn), (c, c)]
    blocks += [(np.full(128, 32767, np.int16), np.full(128, -32768, np.int16))]
    n_ch = len(blocks)
    I = np.stack([np.tile(b[0], (n_blk, 1)) for b in blocks]); Q = np.stack([np.tile(b[1], (n_blk, 1)) for b in blocks])
    b = gpu.AudioSDRpreProcessorBatch(n_ch)
    orcs = [ao.OraclePreProcessor() for _ in range(n_ch)]
    b.startAutoI2SerrorDetection()
    for o in orcs:
        o.startAutoI2SerrorDetection()
    lines = set()
    for k in range(n_blk):
        gi, gq = b.update(I[:, k:k + 1], Q[:, k:k + 1])
        for c, o in enumerate(orcs):
            wi, wq = o.update(I[c, k], Q[c, k])
            assert np.array_equal(gi[c, 0], wi) and np.array_equal(gq[c, 0], wq), (k, c)
            lines.add(o.state()["max_line"])
        _cmp_state(b, orcs)
    assert {0, 5, 122} <= lines, sorted(lines)        # silence -> 0; flat spectrum -> the first scanned line; the last scanned line
    b.close()


def test_pre_autodetect_many_blocks_per_call_and_self_switch_off(gpu, ao):
    """1,100 blocks in calls of 275: the correction changes INSIDE a call and applies from the next block; clean
    channels count 1,001 successes and switch their detector off mid-call; a setter between calls is honoured."""
    n_ch, n_blk, T = 6, 1100, 275
    I, Q = _streams(n_ch, n_blk, [0, 1, 2], seed=8)
    b = gpu.AudioSDRpreProcessorBatch(n_ch)
    orcs = [ao.OraclePreProcessor() for _ in range(n_ch)]
    b.startAutoI2SerrorDetection()
    for o in orcs:
        o.startAutoI2SerrorDetection()
    for k in range(0, n_blk, T):
        if k == 2 * T:
            b.swapIQ(True, ch=4); orcs[4].swapIQ(True)
        gi, gq = b.update(I[:, k:k + T], Q[:, k:k + T])
        for c, o in enumerate(orcs):
            wi, wq = o.update(I[c, k:k + T], Q[c, k:k + T])
            assert np.array_equal(gi[c].reshape(-1), wi) and np.array_equal(gq[c].reshape(-1), wq), (k, c)
        _cmp_state(b, orcs)
    assert b.getAutoI2SerrorDetectionStatus(0) == 0 and b.read_state()["success_count"][0] == 1001
    b.close()


def test_pre_more_channel_quads_than_resident_waves(gpu, ao):
    """The pre-processor's waves are persistent: with more channel quads than the device holds waves a wave walks over several quads with
    the next quad's state and block requested ahead.  32,782 channels (two full rounds of 4,096 waves of four channels + a partial
    one + a partial last quad), 12 distinct streams tiled over them (clean, skewed both ways, noise), detector on, calls of 1 + 3 + 1
    blocks, in place: the 12 distinct ones equal the oracle (audio and state), every channel equals its duplicate."""
    n_ch, uniq, n_blk = 4096 * 4 * 2 + 4 * 3 + 2, 12, 5
    I, Q = _streams(uniq, n_blk, [0, 1, 2, 3], seed=21)
    reps = (n_ch + uniq - 1) // uniq
    It = np.ascontiguousarray(np.tile(I, (reps, 1, 1))[:n_ch]); Qt = np.ascontiguousarray(np.tile(Q, (reps, 1, 1))[:n_ch])
    b = gpu.AudioSDRpreProcessorBatch(n_ch)
    orcs = [ao.OraclePreProcessor() for _ in range(uniq)]
    b.startAutoI2SerrorDetection(); b.setI2SerrorCompensation(1)
    for o in orcs:
        o.startAutoI2SerrorDetection(); o.setI2SerrorCompensation(1)
    gi, gq, k0 = [], [], 0
    for T in (1, 3, 1):
        a_i, a_q = b.update(It[:, k0:k0 + T], Qt[:, k0:k0 + T]); gi.append(a_i); gq.append(a_q); k0 += T
    gi = np.concatenate(gi, axis=1); gq = np.concatenate(gq, axis=1)
    for c, o in enumerate(orcs):
        wi, wq = o.update(I[c], Q[c])
        assert np.array_equal(gi[c].reshape(-1), wi) and np.array_equal(gq[c].reshape(-1), wq), c
    _cmp_state(b, orcs)
    ref_i = np.tile(gi[:uniq], (reps, 1, 1))[:n_ch]; ref_q = np.tile(gq[:uniq], (reps, 1, 1))[:n_ch]
    bad = np.nonzero((gi != ref_i).any(axis=(1, 2)) | (gq != ref_q).any(axis=(1, 2)))[0]
    assert bad.size == 0, "channels differ from their duplicates: %s" % bad[:10]
    st = b.read_state()
    for k in ("correction", "saved_sample", "failure_count", "success_count", "max_line"):
        v = np.asarray(st[k]); assert np.array_equal(v, np.tile(v[:uniq], reps)[:n_ch]), k
    b.close()


def test_pre_device_pointers_in_place_and_strided(gpu, ao):
    """Caller-owned HBM: in place (out == in, as the reference rewrites its blocks) and out of place with strides."""
    n_ch, n_blk = 9, 6
    I, Q = _streams(n_ch, n_blk, [1, 0, 3], seed=5)
    hip = Hip()
    for in_place in (True, False):
        b = gpu.AudioSDRpreProcessorBatch(n_ch)
        orcs = [ao.OraclePreProcessor() for _ in range(n_ch)]
        b.setI2SerrorCompensation(1); b.swapIQ(True, ch=2)
        for c, o in enumerate(orcs):
            o.setI2SerrorCompensation(1)
            if c == 2:
                o.swapIQ(True)
        dI, dQ = hip.upload(I), hip.upload(Q)
        if in_place:
            b.update_device(dI + 256, dQ + 256, dI + 256, dQ + 256, 4, n_blk, n_blk)     # blocks 1..4 of 6
            b.synchronize()
            gi, gq = hip.download(dI, I.shape, np.int16), hip.download(dQ, Q.shape, np.int16)
            assert np.array_equal(gi[:, 0], I[:, 0]) and np.array_equal(gi[:, 5], I[:, 5])
            gi, gq = gi[:, 1:5], gq[:, 1:5]
        else:
            dOi, dOq = hip.malloc(n_ch * 8 * 256), hip.malloc(n_ch * 8 * 256)
            b.update_device(dI + 256, dQ + 256, dOi + 512, dOq + 512, 4, n_blk, 8)
            b.synchronize()
            gi = hip.download(dOi, (n_ch, 8, 128), np.int16)[:, 2:6]
            gq = hip.download(dOq, (n_ch, 8, 128), np.int16)[:, 2:6]
            assert np.array_equal(hip.download(dI, I.shape, np.int16), I)                # inputs untouched
        for c, o in enumerate(orcs):
            wi, wq = o.update(I[c, 1:5], Q[c, 1:5])
            assert np.array_equal(gi[c].reshape(-1), wi) and np.array_equal(gq[c].reshape(-1), wq), (in_place, c)
        assert b.last_kernel_ms() > 0
        b.update_device(0, dQ, dI, dQ, 1)                                                # missing input: no-op
        _cmp_state(b, orcs)
        with pytest.raises(gpu.AsdrError, match="aligned"):
            b.update_device(dI + 2, dQ, dI, dQ, 1)
        b.close()
    hip.free_all()


@pytest.mark.parametrize("calls", [(4, 1, 1, 1), (1, 2, 3, 1), (2, 5), (1, 1, 1, 1, 1, 1, 1)])
@pytest.mark.parametrize("n_ch", [1, 7, 8, 9, 40])
def test_iqgen_parity(gpu, ao, n_ch, calls):
    """AudioIQgenerator: 7 blocks as calls of 4 + 1 + 1 + 1 blocks (and other splits: the two carried blocks live in a raw int16 ring whose
    slots a call of any length must leave right), per-channel gain balance; int16 bit-exact."""
    n_blk = 7
    rng = np.random.default_rng(n_ch)
    t = np.arange(n_blk * 128)
    x = np.stack([np.clip(9000 * np.sin(2 * np.pi * (700.0 + 613.0 * c) * t / FS) + 3000 * rng.standard_normal(t.size), -32768, 32767)
                  for c in range(n_ch)]).astype(np.int16).reshape(n_ch, n_blk, 128)
    if n_ch > 2:
        x[2] = 32767                                       # full scale with gain 4: int32-saturate-then-truncate convention
    g = gpu.AudioIQgeneratorBatch(n_ch)
    orcs = [ao.OracleIQgenerator() for _ in range(n_ch)]
    for c in range(n_ch):
        if c % 3:
            bal = (1.0, 1.02, 4.0)[c % 3]
            g.setGainBalance(bal, ch=c); orcs[c].setGainBalance(bal)
    want = [o.update(x[c]) for c, o in enumerate(orcs)]
    parts, k0 = [], 0
    for T in calls:
        parts.append(g.update(x[:, k0:k0 + T])); k0 += T
    gi = np.concatenate([p[0] for p in parts], axis=1); gq = np.concatenate([p[1] for p in parts], axis=1)
    for c in range(n_ch):
        assert np.array_equal(gi[c].reshape(-1), want[c][0]), "I ch %d" % c
        assert np.array_equal(gq[c].reshape(-1), want[c][1]), "Q ch %d" % c
    assert g.last_kernel_ms() > 0
    g.close()


def test_iqgen_more_channel_groups_than_resident_waves(gpu, ao):
    """The IQ generator's waves are persistent: with more channel groups than the device holds waves, a wave walks over several groups
    with the next group's rows requested ahead.  49,195 channels (2 full rounds of 3,072 waves + a partial one + a partial last group),
    64 distinct inputs tiled over them: every channel equals its duplicate, the 64 distinct ones equal the oracle; calls of 1 + 2 + 1 blocks."""
    n_ch, uniq, n_blk = 3072 * 8 * 2 + 8 * 5 + 3, 64, 4
    rng = np.random.default_rng(11)
    t = np.arange(n_blk * 128)
    x = np.stack([np.clip(9000 * np.sin(2 * np.pi * (500.0 + 97.0 * c) * t / FS) + 2500 * rng.standard_normal(t.size), -32768, 32767)
                  for c in range(uniq)]).astype(np.int16).reshape(uniq, n_blk, 128)
    reps = (n_ch + uniq - 1) // uniq
    xt = np.ascontiguousarray(np.tile(x, (reps, 1, 1))[:n_ch])
    g = gpu.AudioIQgeneratorBatch(n_ch)
    orcs = [ao.OracleIQgenerator() for _ in range(uniq)]
    for c in range(uniq):
        if c % 3:
            bal = (1.0, 1.02, 0.97)[c % 3]
            orcs[c].setGainBalance(bal)
            for d in range(c, n_ch, uniq):
                g.setGainBalance(bal, ch=d)
    want = [o.update(x[c]) for c, o in enumerate(orcs)]
    parts, k0 = [], 0
    for T in (1, 2, 1):
        parts.append(g.update(xt[:, k0:k0 + T])); k0 += T
    gi = np.concatenate([p[0] for p in parts], axis=1); gq = np.concatenate([p[1] for p in parts], axis=1)
    for c in range(uniq):
        assert np.array_equal(gi[c].reshape(-1), want[c][0]), "I ch %d" % c
        assert np.array_equal(gq[c].reshape(-1), want[c][1]), "Q ch %d" % c
    ref_i = np.tile(gi[:uniq], (reps, 1, 1))[:n_ch]; ref_q = np.tile(gq[:uniq], (reps, 1, 1))[:n_ch]
    bad = np.nonzero((gi != ref_i).any(axis=(1, 2)) | (gq != ref_q).any(axis=(1, 2)))[0]
    assert bad.size == 0, "channels differ from their duplicates: %s" % bad[:10]
    g.close()


def test_grabber_parity(gpu, ao):
    """AudioGrabberComplex256: every call pattern of odd/even block counts against the oracle's protocol."""
    n_ch = 5
    rng = np.random.default_rng(2)
    I = rng.integers(-32768, 32767, size=(n_ch, 16, 128)).astype(np.int16)
    Q = rng.integers(-32768, 32767, size=(n_ch, 16, 128)).astype(np.int16)
    g = gpu.AudioGrabberComplex256Batch(n_ch)
    orcs = [ao.OracleGrabber() for _ in range(n_ch)]
    d = np.full(512, 77, dtype=np.int16)
    copied, _ = g.grab(1, d)
    assert copied == 0 and (d == 77).all()
    pos = 0
    for T in (1, 1, 1, 2, 3, 2, 1, 4, 1):
        g.update(I[:, pos:pos + T], Q[:, pos:pos + T])
        for c, o in enumerate(orcs):
            o.update(I[c, pos:pos + T], Q[c, pos:pos + T])
        pos += T
        for c, o in enumerate(orcs):
            assert g.newDataAvailable(c) == o.newDataAvailable(), (pos, c)
        c = pos % n_ch                                     # grab one channel per step: flags diverge per channel
        want = orcs[c].grab(np.full(512, 77, dtype=np.int16))
        copied, got = g.grab(c, np.full(512, 77, dtype=np.int16))
        assert np.array_equal(got, want), (pos, c)
        assert g.newDataAvailable(c) == 0
    copied, allbuf = g.grab_all()
    assert copied == 1
    for c, o in enumerate(orcs):
        assert np.array_equal(allbuf[c], o.grab())
        assert g.newDataAvailable(c) == 0
    hip = Hip()
    assert np.array_equal(hip.download(g.device_ptr(), (n_ch, 512), np.int16), allbuf)
    g.close()


def test_front_end_chain_on_device(gpu, ao):
    """IQ generator -> pre-processor -> AudioSDR -> capture sink, all on HBM buffers on one stream, against the same
    chain of oracles (the documented audio graph of EXTRAS/BareBonesWSPR with a synthetic source)."""
    n_ch, n_blk = 12, 24
    t = np.arange(n_blk * 128)
    rng = np.random.default_rng(4)
    x = np.stack([6000 * np.sin(2 * np.pi * (6890.0 - 1500.0 + 20.0 * c) * t / FS) + 800 * rng.standard_normal(t.size)
                  for c in range(n_ch)]).astype(np.int16).reshape(n_ch, n_blk, 128)
    gen, pre, sdr = gpu.AudioIQgeneratorBatch(n_ch), gpu.AudioSDRpreProcessorBatch(n_ch), gpu.AudioSDRBatch(n_ch)
    pre.startAutoI2SerrorDetection()
    sdr.setDemodMode(gpu.WSPRmode); sdr.setAudioFilter(gpu.audioWSPR); sdr.enableAudioFilter(); sdr.setAGCmode(gpu.AGCmedium)
    hip = Hip()
    s = hip.stream()
    dX = hip.upload(x)
    dI, dQ = hip.malloc(x.nbytes), hip.malloc(x.nbytes)
    sdr.capture_open(n_blk)
    for k in range(0, n_blk, 8):
        off = k * 256
        gen.update_device(dX + off, dI + off, dQ + off, 8, n_blk, n_blk, stream=s)
        pre.update_device(dI + off, dQ + off, dI + off, dQ + off, 8, n_blk, n_blk, stream=s)
        sdr.capture_update_device(dI + off, dQ + off, 8, in_stride_blocks=n_blk, stream=s)
    hip.sync(s)
    for c in range(n_ch):
        og, op, osd = ao.OracleIQgenerator(), ao.OraclePreProcessor(), ao.OracleSDR()
        op.startAutoI2SerrorDetection()
        osd.setDemodMode(ao.WSPRmode); osd.setAudioFilter(ao.audioWSPR); osd.enableAudioFilter(); osd.setAGCmode(ao.AGCmedium)
        i, q = og.update(x[c])
        i, q = op.update(i, q)
        want = osd.update(i.reshape(n_blk, 128), q.reshape(n_blk, 128))
        assert np.array_equal(sdr.capture_read(c), want), "ch %d" % c
    hip.free_all()
    for o in (gen, pre, sdr):
        o.close()


def test_grabber_power_spectrum_parity(gpu, ao):
    """Device-side panadapter: |FFT256|^2 of every channel's grabber buffer, float32 bit-exact against the oracle's FFT."""
    n_ch = 7
    rng = np.random.default_rng(6)
    t = np.arange(256)
    I = np.stack([12000 * np.cos(2 * np.pi * (3 + 11 * c) * t / 256) + 300 * rng.standard_normal(256) for c in range(n_ch)]).astype(np.int16)
    Q = np.stack([12000 * np.sin(2 * np.pi * (3 + 11 * c) * t / 256) + 300 * rng.standard_normal(256) for c in range(n_ch)]).astype(np.int16)
    g = gpu.AudioGrabberComplex256Batch(n_ch)
    valid, spec = g.power_spectrum()
    assert valid == 0 and not spec.any()                       # no complete buffer yet: destination untouched
    g.update(I.reshape(n_ch, 2, 128), Q.reshape(n_ch, 2, 128))
    valid, spec = g.power_spectrum()
    assert valid == 1
    _, bufs = g.grab_all()
    for c in range(n_ch):
        want = ao.grab_power_spectrum(bufs[c])
        assert np.array_equal(f32_bits(spec[c]), f32_bits(want)), "ch %d" % c
        assert int(np.argmax(spec[c])) == 3 + 11 * c
    hip = Hip()
    dS = hip.malloc(n_ch * 256 * 4)
    assert g.power_spectrum_device(dS) == 1
    g.synchronize(); hip.sync()
    assert np.array_equal(f32_bits(hip.download(dS, (n_ch, 256), np.float32)), f32_bits(spec))
    hip.free_all(); g.close()
