"""GPU parity tests proper: every call goes through the C ABI of libasdr_hip.so (ctypes), results are compared
with the CPU oracle on the same seeded inputs and with the committed fixtures.

Bar (north_star): bit-exact for mode/index logic; stated float32 tolerance for the signal path.  The HIP path keeps
the reference's operation order with FMA contraction off and evaluates the reference's double-precision islands in
binary64, so the tolerance asserted here is ZERO: int16 outputs, every float32 stage tap and every status getter
must be bit-identical to the oracle."""
import os

import numpy as np
import pytest

from cases import CASES
from helpers import S, apply_setters, compare_status, f32_bits

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "chain_vectors.npz")


def _mk(gpu, ao, n_ch, setters, taps=False):
    batch = gpu.AudioSDRBatch(n_ch)
    if taps:
        batch.enable_taps(True)
    orcs = [ao.OracleSDR(taps=taps) for _ in range(n_ch)]
    apply_setters(batch, orcs, setters)
    return batch, orcs


@pytest.mark.parametrize("name", sorted(CASES))
def test_case_block_by_block_with_taps(gpu, ao, name):
    """One update() per block; compares int16 audio, all 12 float32 stage taps of every block, and status."""
    from audiosdr_amd.synth import make_iq
    n_ch, n_blk, setters, sig = CASES[name]
    I, Q = make_iq(n_ch, n_blk, **sig)
    batch, orcs = _mk(gpu, ao, n_ch, setters, taps=True)
    for b in range(n_blk):
        got = batch.update(I[:, b:b + 1], Q[:, b:b + 1])[:, 0]
        taps = batch.read_taps()
        for c in range(n_ch):
            want = orcs[c].update(I[c, b], Q[c, b])
            for t in gpu.TAPS:
                assert np.array_equal(f32_bits(taps[t][c]), f32_bits(orcs[c].tap(t))), "%s block %d ch %d tap %s" % (name, b, c, t)
            assert np.array_equal(got[c], want), "%s block %d ch %d" % (name, b, c)
    compare_status(gpu, batch, orcs)
    batch.close()


@pytest.mark.parametrize("name", sorted(CASES))
def test_case_multi_block_call_matches_golden(gpu, name):
    """All blocks in ONE asdr_update() call (in-kernel block loop) against the committed fixtures -- no oracle."""
    from audiosdr_amd.synth import make_iq
    g = np.load(GOLD)
    n_ch, n_blk, setters, sig = CASES[name]
    I, Q = make_iq(n_ch, n_blk, **sig)
    batch = gpu.AudioSDRBatch(n_ch)
    for meth, args, sel in setters:
        for c in range(n_ch):
            if sel is None or sel(c):
                getattr(batch, meth)(*args, ch=c)
    got = batch.update(I, Q)
    assert np.array_equal(got, g[name + "/out"])
    st = batch.read_status()
    assert np.array_equal(np.stack([st["agc_active"], st["nb_detected"], st["sam_locked"]], axis=1), g[name + "/status"])
    assert np.array_equal(f32_bits(np.stack([st["sam_frequency"], st["am_carrier"]], axis=1)), f32_bits(g[name + "/fstatus"]))
    batch.close()


@pytest.mark.parametrize("n_ch", [1, 5, 7, 8, 9, 17, 64, 67])
def test_ragged_channel_counts(gpu, ao, n_ch):
    """Waves hold 8 channels; counts that are not multiples of 8 use padded (dummy) slots."""
    from audiosdr_amd.synth import make_iq
    I, Q = make_iq(n_ch, 5, fc=6290.0, A=0.25, impulse_every=500)
    setters = [S("setDemodMode", 1), S("enableAudioFilter")]
    batch, orcs = _mk(gpu, ao, n_ch, setters)
    got = batch.update(I, Q)
    want = np.stack([orcs[c].update(I[c], Q[c]).reshape(5, 128) for c in range(n_ch)])
    assert np.array_equal(got, want)
    batch.close()


def test_batched_equals_independent_single_channel_batches(gpu):
    """an N-channel batch == N one-channel batches, bit for bit (channels never interact)."""
    from audiosdr_amd.synth import make_iq
    n_ch = 19
    I, Q = make_iq(n_ch, 6, fc=6890.0, A=0.3, m=0.5)
    big = gpu.AudioSDRBatch(n_ch)
    for c in range(n_ch):
        big.setDemodMode(c % 7, ch=c)
    big.setNoiseBlankerThresholdDb(10.0)
    got = big.update(I, Q)
    for c in range(n_ch):
        one = gpu.AudioSDRBatch(1)
        one.setDemodMode(c % 7)
        one.setNoiseBlankerThresholdDb(10.0)
        assert np.array_equal(one.update(I[c:c + 1], Q[c:c + 1])[0], got[c]), c
        one.close()
    big.close()


def test_setters_between_blocks(gpu, ao):
    """Control-plane calls between update()s: mode switches zero the IF filter state (AudioSDR.cpp:191-218) but keep
    the Hilbert history and mixer phases; setAudioFilter zeroes the audio filter state (:300-309); NB setters and
    enableNoiseBlanker reset the blanker (:653-674); enableALSfilter zeroes taps+history (:384-391); init() (:174-185)."""
    from audiosdr_amd.synth import make_iq
    n_ch, n_blk = 4, 24
    I, Q = make_iq(n_ch, n_blk, fc=6600.0, A=0.3, m=0.4, impulse_every=777, f2=7300.0, a2=0.1)
    batch, orcs = _mk(gpu, ao, n_ch, [S("setDemodMode", 1), S("enableAudioFilter")])
    script = {
        3: [S("setDemodMode", 0)], 5: [S("setDemodMode", 4)], 7: [S("setDemodMode", 5), S("setNoiseBlankerThresholdDb", 10.0)],
        9: [S("setAudioFilter", 0)], 10: [S("disableNoiseBlanker")], 12: [S("enableNoiseBlanker")], 13: [S("setDemodMode", 6)],
        14: [S("enableALSfilter")], 16: [S("setALSfilterPeak")], 17: [S("enableALSfilter"), S("setALSfilterNotch")],
        18: [S("setAGCmode", 1), S("setOutputGain", 0.8)], 19: [S("init")], 20: [S("setMute", 1)], 21: [S("setMute", 0), S("setDemodMode", 3)],
        22: [S("setDemodMode", 1, sel=lambda c: c % 2 == 0), S("setInputGain", 2.0, sel=lambda c: c == 3)],
    }
    for b in range(n_blk):
        if b in script:
            apply_setters(batch, orcs, script[b])
        got = batch.update(I[:, b:b + 1], Q[:, b:b + 1])[:, 0]
        for c in range(n_ch):
            assert np.array_equal(got[c], orcs[c].update(I[c, b], Q[c, b])), "block %d ch %d" % (b, c)
    compare_status(gpu, batch, orcs)
    batch.close()


def test_mixer_uniform_and_divergent_phases_in_one_wave(gpu, ao):
    """The mixer computes the phase sequence and its sin/cos once per wave when all 8 channels of the wave carry the same
    (phase, increment) and falls back to the per-channel path otherwise (AudioSDR.h:508-526).  16 channels start in two
    different SSB modes (different phase increments), then all switch to USB: equal schedule keys put them in the same
    waves with DIFFERENT carried phases.  A second batch keeps all 16 in USB from the start (uniform waves)."""
    from audiosdr_amd.synth import make_iq
    n_ch, n_blk = 16, 7
    I, Q = make_iq(n_ch, n_blk, fc=6290.0, A=0.25, impulse_every=600)
    for divergent in (True, False):
        pre = [S("setDemodMode", 0, sel=lambda c: c % 2 == 0), S("setDemodMode", 1, sel=lambda c: c % 2 == 1)] if divergent else [S("setDemodMode", 1)]
        batch, orcs = _mk(gpu, ao, n_ch, pre + [S("enableAudioFilter")], taps=True)
        for b in range(n_blk):
            if b == 2:
                apply_setters(batch, orcs, [S("setDemodMode", 1)])
            got = batch.update(I[:, b:b + 1], Q[:, b:b + 1])[:, 0]
            taps = batch.read_taps()
            for c in range(n_ch):
                want = orcs[c].update(I[c, b], Q[c, b])
                for t in ("MIX_I", "MIX_Q"):
                    assert np.array_equal(f32_bits(taps[t][c]), f32_bits(orcs[c].tap(t))), "divergent=%s block %d ch %d tap %s" % (divergent, b, c, t)
                assert np.array_equal(got[c], want), "divergent=%s block %d ch %d" % (divergent, b, c)
        batch.close()


@pytest.mark.parametrize("mode", [7, -1, 65535, 9])
@pytest.mark.parametrize("als", [False, True])
def test_unknown_mode_values(gpu, ao, mode, als):
    """setDemodMode with a value outside 0..6 (AudioSDR.cpp:188: only _mode changes).  Neither demodulator branch runs
    (.cpp:84, 122), _audioOut keeps what the previous block left in it and the audio filter / AGC / ALS / output stage process
    it AGAIN (.cpp:149-161) -- block after block while the mode stays unknown.  The product keeps every block's post-ALS row in
    HBM for this (asdr_device.h audio_prev); the oracle is the DEFAULT one (the reference's behaviour, no product-modelling
    switch).  Audio filter + AGC on, with and without the ALS filter (its kernel kinds), per-channel and whole-batch switches,
    a multi-block call inside the unknown stretch, returns to known modes; int16 audio, six stage taps, status."""
    from audiosdr_amd.synth import make_iq
    n_ch, n_blk = 19, 16
    I, Q = make_iq(n_ch, n_blk, fc=6290.0, A=0.25, impulse_every=700, f2=7100.0, a2=0.1)
    pre = [S("setDemodMode", 1), S("enableAudioFilter")] + ([S("enableALSfilter")] if als else [])
    batch, orcs = _mk(gpu, ao, n_ch, pre, taps=True)
    script = {3: [S("setDemodMode", mode, sel=lambda c: c % 2 == 0)], 6: [S("setDemodMode", 4, sel=lambda c: c % 4 == 0)],
              8: [S("setDemodMode", mode)], 13: [S("setDemodMode", 0)], 14: [S("setDemodMode", mode, sel=lambda c: c < 9), S("setMute", 1, sel=lambda c: c == 2)]}
    calls = [1] * 9 + [3] + [1] * 4      # blocks per update(): blocks 9..11 in ONE call (the in-kernel block loop keeps re-processing)
    b = 0
    for nb in calls:
        if b in script:
            apply_setters(batch, orcs, script[b])
        got = batch.update(I[:, b:b + nb], Q[:, b:b + nb])
        taps = batch.read_taps()
        for c in range(n_ch):
            want = orcs[c].update(I[c, b:b + nb], Q[c, b:b + nb]).reshape(nb, 128)
            for t in ("NB_I", "IF_I", "IF_Q", "DEMOD", "AUDIO_FILT", "AGC", "ALS"):
                assert np.array_equal(f32_bits(taps[t][c]), f32_bits(orcs[c].tap(t))), "mode %d block %d ch %d tap %s" % (mode, b, c, t)
            assert np.array_equal(got[c], want), "mode %d block %d ch %d" % (mode, b, c)
            assert batch.getDemodMode(ch=c) == orcs[c].getDemodMode()
        b += nb
    assert b == n_blk
    compare_status(gpu, batch, orcs)
    batch.close()


def test_unknown_mode_values_without_the_kept_row(gpu, ao):
    """asdr_set_exact_unknown_mode(b, 0): the post-ALS row is not kept (512 B per channel-block less written) and an unknown mode value
    processes a SILENT block -- the oracle's ao_set_unknown_mode_silence models that opt-out; everything else stays bit-exact.
    Switching the row back on starts from silence and is exact again from the first block with a known mode."""
    from audiosdr_amd.synth import make_iq
    n_ch, n_blk = 11, 14
    I, Q = make_iq(n_ch, n_blk, fc=6290.0, A=0.25, impulse_every=700)
    batch, orcs = _mk(gpu, ao, n_ch, [S("setDemodMode", 1), S("enableAudioFilter")], taps=True)
    batch.set_exact_unknown_mode(False)
    for o in orcs:
        o.set_unknown_mode_silence()
    script = {3: [S("setDemodMode", 7, sel=lambda c: c % 2 == 0)], 6: [S("setDemodMode", 4, sel=lambda c: c % 4 == 0)],
              8: [S("setDemodMode", -1)], 10: [S("setDemodMode", 0)], 12: [S("setDemodMode", 9, sel=lambda c: c % 3 == 0)]}
    for b in range(n_blk):
        if b in script:
            apply_setters(batch, orcs, script[b])
        if b == 10:   # known modes again: keep the row from here on
            batch.set_exact_unknown_mode(True)
            for o in orcs:
                o.set_unknown_mode_silence(False)
        got = batch.update(I[:, b:b + 1], Q[:, b:b + 1])[:, 0]
        taps = batch.read_taps()
        for c in range(n_ch):
            want = orcs[c].update(I[c, b], Q[c, b])
            for t in ("NB_I", "IF_I", "IF_Q", "DEMOD", "AUDIO_FILT", "AGC"):
                assert np.array_equal(f32_bits(taps[t][c]), f32_bits(orcs[c].tap(t))), "block %d ch %d tap %s" % (b, c, t)
            assert np.array_equal(got[c], want), "block %d ch %d" % (b, c)
    compare_status(gpu, batch, orcs)
    batch.close()


def test_local_oscillator_cache_hits_misses_and_multi_block_calls(gpu, ao):
    """The mixer's sin/cos pairs of the next block are left in HBM by wave 0 and read by every wave whose channels start that
    block with exactly the cached (phase, increment) (asdr_device.h LoEntry).  Sequence: uniform USB batch (misses once, then
    hits), a multi-block call (only its first block can hit), all channels to LSB (new increment: miss, then hits), half of the
    channels to CW (two key groups with different increments: wave 0's group hits, the other computes its own), back to one
    group.  Every block of every channel against the oracle."""
    from audiosdr_amd.synth import make_iq
    n_ch = 40                                       # 5 waves
    plan = [1, 1, 1, 3, 1, 1, 2, 1, 1, 1, 1, 4, 1]  # blocks per update() call
    script = {4: [S("setDemodMode", 0)], 7: [S("setDemodMode", 3, sel=lambda c: c >= 16)], 10: [S("setDemodMode", 1)]}
    I, Q = make_iq(n_ch, sum(plan), fc=6290.0, A=0.25, impulse_every=900)
    batch, orcs = _mk(gpu, ao, n_ch, [S("setDemodMode", 1), S("enableAudioFilter")])
    b0 = 0
    for call, nb in enumerate(plan):
        if call in script:
            apply_setters(batch, orcs, script[call])
        got = batch.update(I[:, b0:b0 + nb], Q[:, b0:b0 + nb])
        for c in range(n_ch):
            want = orcs[c].update(I[c, b0:b0 + nb], Q[c, b0:b0 + nb]).reshape(nb, 128)
            assert np.array_equal(got[c], want), "call %d (blocks %d..%d) ch %d" % (call, b0, b0 + nb - 1, c)
        b0 += nb
    compare_status(gpu, batch, orcs)
    batch.close()


def test_agc_hanging_chunks_and_attacks(gpu, ao):
    """The AGC recurrence has three wave-uniform forms per 8-sample chunk: (a) no sample attacks and the hang counter cannot run
    out -> only the counter moves; (b) the counter cannot run out (counter >= 8, hang >= 8) -> only attacking samples are
    evaluated; (c) the general per-sample form (release possible).  Waves 0 / 1 / 2 are homogeneous so that (a) and (b) are
    taken (long hang: USB; tiny or zero hang: always (c); AM, whose |x| is the constant carrier level), wave 3 mixes all of
    them.  Amplitude steps up and down drive attacks, pure hang, the counter running out and release; AGC tap + int16 output."""
    from audiosdr_amd.synth import make_iq
    n_ch, n_blk = 40, 40            # (wave 4: zero hang time in every channel -- the every-sample-updates form of (c))
    I, Q = make_iq(n_ch, n_blk, fc=6290.0, A=0.3, noise=0.002)
    env = np.ones(n_blk * 128)
    env[128 * 6:128 * 12] = 0.05; env[128 * 12:128 * 13] = 1.0; env[128 * 13:128 * 30] = 0.02; env[128 * 30:] = 0.6
    I = (I.reshape(n_ch, -1) * env).astype(np.int16).reshape(n_ch, n_blk, 128)
    Q = (Q.reshape(n_ch, -1) * env).astype(np.int16).reshape(n_ch, n_blk, 128)
    grp = lambda c: (c // 8) if c < 24 else ((c % 3) if c < 32 else 3)
    setters = [S("setDemodMode", 1), S("disableNoiseBlanker"),
               S("setAGChangTime", 7.0, sel=lambda c: grp(c) == 0 and c % 2 == 0),      # 308 samples: runs out inside chunks
               S("setAGCmode", 3, sel=lambda c: grp(c) == 0 and c % 4 == 1),
               S("setAGChangTime", 0.1, sel=lambda c: grp(c) == 1 and c % 2 == 0),      # 4 samples: below the chunk length
               S("setAGChangTime", 0.0, sel=lambda c: grp(c) == 1 and c % 2 == 1),
               S("setAGChangTime", 0.0, sel=lambda c: grp(c) == 3),
               S("setAGCmode", 1, sel=lambda c: grp(c) == 3 and c % 2 == 1),
               S("setDemodMode", 4, sel=lambda c: grp(c) == 2),
               S("setAGCmode", 1, sel=lambda c: grp(c) == 2 and c % 2 == 1)]
    batch, orcs = _mk(gpu, ao, n_ch, setters, taps=True)
    for b in range(n_blk):
        got = batch.update(I[:, b:b + 1], Q[:, b:b + 1])[:, 0]
        taps = batch.read_taps()
        for c in range(n_ch):
            want = orcs[c].update(I[c, b], Q[c, b])
            assert np.array_equal(f32_bits(taps["AGC"][c]), f32_bits(orcs[c].tap("AGC"))), "block %d ch %d AGC tap" % (b, c)
            assert np.array_equal(got[c], want), "block %d ch %d" % (b, c)
    compare_status(gpu, batch, orcs)
    batch.close()


def test_agc_quiet_blocks(gpu, ao):
    """The AGC's block-level fast form: when in NO channel of a wave a sample exceeds the envelope and no hang counter can run out
    inside the block (counter >= 128, hang >= 8), the chunk loop, the gain table and the per-sample rows are skipped -- counters drop
    by 128, every sample takes the gain carried in.  A loud stretch, then a soft one: every channel hangs until its counter runs
    out.  Hang times swept around one block (2.6 .. 6.2 ms = 114 .. 273 samples) put the counters on either side of 128 at the
    block boundaries (some channels of a wave quiet, others not: the wave takes the chunk loop); wave 0 keeps the default hang
    time (4,410 samples: ~34 quiet blocks in a row, then release and attacks), wave 2 is AM (|x| = the carrier level), wave 3 has
    the blanker on and the audio filter off.  AGC tap, int16 output and the status bits, block by block."""
    from audiosdr_amd.synth import make_iq
    n_ch, n_blk = 32, 64
    I, Q = make_iq(n_ch, n_blk, fc=6290.0, A=0.3, noise=0.001)
    env = np.ones(n_blk * 128)
    env[128 * 8:128 * 50] = 0.3; env[128 * 50:128 * 52] = 1.0; env[128 * 52:] = 0.1
    I = (I.reshape(n_ch, -1) * env).astype(np.int16).reshape(n_ch, n_blk, 128)
    Q = (Q.reshape(n_ch, -1) * env).astype(np.int16).reshape(n_ch, n_blk, 128)
    setters = [S("setDemodMode", 1), S("enableAudioFilter", sel=lambda c: c < 24), S("disableNoiseBlanker", sel=lambda c: c < 24)]
    setters += [S("setAGChangTime", 2.6 + 0.45 * (c - 8), sel=(lambda k, c=c: k == c)) for c in range(8, 16)]
    setters += [S("setDemodMode", 4, sel=lambda c: 16 <= c < 24), S("setAGChangTime", 3.1, sel=lambda c: c in (17, 21, 26, 29))]
    batch, orcs = _mk(gpu, ao, n_ch, setters, taps=True)
    for b in range(n_blk):
        got = batch.update(I[:, b:b + 1], Q[:, b:b + 1])[:, 0]
        taps = batch.read_taps()
        for c in range(n_ch):
            want = orcs[c].update(I[c, b], Q[c, b])
            assert np.array_equal(f32_bits(taps["AGC"][c]), f32_bits(orcs[c].tap("AGC"))), "block %d ch %d AGC tap" % (b, c)
            assert np.array_equal(got[c], want), "block %d ch %d" % (b, c)
        if b in (9, 20, 45, 51, 63):
            compare_status(gpu, batch, orcs)
    # the same settings without taps, four blocks per call (the multi-block loop keeps the state in HBM between blocks)
    batch2, orcs2 = _mk(gpu, ao, n_ch, setters)
    for b in range(0, n_blk, 4):
        got = batch2.update(I[:, b:b + 4], Q[:, b:b + 4])
        for c in range(n_ch):
            want = orcs2[c].update(I[c, b:b + 4], Q[c, b:b + 4]).reshape(4, 128)
            assert np.array_equal(got[c], want), "blocks %d.. ch %d" % (b, c)
    compare_status(gpu, batch2, orcs2)
    batch.close(); batch2.close()


def test_calls_alternating_between_two_streams(gpu, ao):
    """asdr_update_device on a different stream than the previous call waits (event) for that call's kernels: every launch
    read-modify-writes the same per-channel state.  Two caller-owned streams used alternately, no host synchronisation in
    between; a batch with three kernel instantiations (plain / SAM / ALS) so the internal helper streams are in play too."""
    from audiosdr_amd.synth import make_iq
    import ctypes as C
    from helpers import Hip
    hip = Hip()
    n_ch, n_blk = 27, 12
    I, Q = make_iq(n_ch, n_blk, fc=6890.0, A=0.3, m=0.5)
    setters = [S("setNoiseBlankerThresholdDb", 10.0)] + [S("setDemodMode", m, sel=(lambda c, m=m: c % 7 == m)) for m in range(7)] + \
              [S("enableALSfilter", sel=lambda c: c % 5 == 0)]
    batch, orcs = _mk(gpu, ao, n_ch, setters)
    s = [C.c_void_p(), C.c_void_p()]
    for x in s:
        assert hip.h.hipStreamCreate(C.byref(x)) == 0
    dI = [hip.upload(I[:, b].copy()) for b in range(n_blk)]
    dQ = [hip.upload(Q[:, b].copy()) for b in range(n_blk)]
    dO = [hip.malloc(n_ch * 256) for _ in range(n_blk)]
    L = gpu.load_library()
    for b in range(n_blk):
        assert L.asdr_update_device(batch._h, C.c_void_p(dI[b]), C.c_void_p(dQ[b]), C.c_void_p(dO[b]), 1, s[b & 1]) == 0
    for x in s:
        hip.h.hipStreamSynchronize(x)
    for b in range(n_blk):
        got = np.zeros((n_ch, 128), np.int16)
        assert hip.h.hipMemcpy(got.ctypes.data_as(C.c_void_p), C.c_void_p(dO[b]), got.nbytes, 2) == 0
        for c in range(n_ch):
            assert np.array_equal(got[c], orcs[c].update(I[c, b], Q[c, b])), "block %d ch %d" % (b, c)
    for x in s:
        hip.h.hipStreamDestroy(x)
    batch.close()


def test_missing_input_guard(gpu, ao):
    """AudioSDR.cpp:48-56: a missing I or Q block -> return without processing; state does not advance."""
    import ctypes as C
    from audiosdr_amd.synth import make_iq
    I, Q = make_iq(2, 4, fc=6290.0, A=0.25)
    batch, orcs = _mk(gpu, ao, 2, [S("setDemodMode", 1)])
    L = gpu.load_library()
    out = np.full((2, 1, 128), 77, np.int16)
    p = C.POINTER(C.c_int16)
    assert L.asdr_update(batch._h, None, Q[:, 0:1].copy().ctypes.data_as(p), out.ctypes.data_as(p), 1) == 0
    assert L.asdr_update(batch._h, I[:, 0:1].copy().ctypes.data_as(p), None, out.ctypes.data_as(p), 1) == 0
    assert (out == 77).all()
    got = batch.update(I, Q)
    want = np.stack([orcs[c].update(I[c], Q[c]).reshape(4, 128) for c in range(2)])
    assert np.array_equal(got, want)
    batch.close()


def test_all_int16_inputs_through_the_scale_stage(gpu, ao):
    """Input scaling is a double-precision island (AudioSDR.cpp:68-69): all 65,536 int16 values, several gains."""
    vals = np.arange(-32768, 32768, dtype=np.int32).astype(np.int16).reshape(512, 1, 128)
    for gain, bal in [(1.0, None), (0.3, 1.02), (10.0, None), (2.7182817, 0.9)]:
        batch = gpu.AudioSDRBatch(512)
        batch.enable_taps(True)
        batch.setInputGain(gain)
        o = ao.OracleSDR(taps=True)
        o.setInputGain(gain)
        if bal is not None:
            batch.setIQgainBalance(bal); o.setIQgainBalance(bal)
        batch.update(vals, vals[::-1].copy())
        taps = batch.read_taps()
        for c in range(0, 512, 37):
            o.update(vals[c, 0], vals[511 - c, 0])
            assert np.array_equal(f32_bits(taps["SCALED_I"][c]), f32_bits(o.tap("SCALED_I")))
            assert np.array_equal(f32_bits(taps["SCALED_Q"][c]), f32_bits(o.tap("SCALED_Q")))
        # and the whole tap against the oracle's scalar helper, every value
        L = ao.lib()
        gi = np.float32(gain) if bal is None else np.float32(np.float32(gain) * np.sqrt(np.float32(bal)))
        want = np.array([np.float32(L.ao_scale_sample(int(v), float(gi))) for v in vals.reshape(-1)], dtype=np.float32)
        assert np.array_equal(f32_bits(taps["SCALED_I"].reshape(-1)), f32_bits(want))
        batch.close()


def test_device_pointer_entry_point_and_stream(gpu, ao):
    """asdr_update_device with caller-owned HBM buffers on a caller-created side stream == host-pointer entry point.
    HBM and the stream come straight from the HIP runtime (ctypes on libamdhip64), as a C host application would do."""
    import ctypes as C
    from audiosdr_amd.synth import make_iq
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]
    hip.hipStreamSynchronize.argtypes = [C.c_void_p]
    hip.hipFree.argtypes = [C.c_void_p]
    n_ch, n_blk = 33, 4
    I, Q = make_iq(n_ch, n_blk, fc=6290.0, A=0.25)
    batch, orcs = _mk(gpu, ao, n_ch, [S("setDemodMode", 1), S("enableAudioFilter")])
    nbytes = I.nbytes
    dI, dQ, dO, stream = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
    assert hip.hipMalloc(C.byref(dI), nbytes) == 0 and hip.hipMalloc(C.byref(dQ), nbytes) == 0 and hip.hipMalloc(C.byref(dO), nbytes) == 0
    assert hip.hipStreamCreate(C.byref(stream)) == 0
    H2D, D2H = 1, 2
    assert hip.hipMemcpy(dI, I.ctypes.data_as(C.c_void_p), nbytes, H2D) == 0
    assert hip.hipMemcpy(dQ, Q.ctypes.data_as(C.c_void_p), nbytes, H2D) == 0
    batch.update_device(dI.value, dQ.value, dO.value, n_blk, stream.value)
    assert hip.hipStreamSynchronize(stream) == 0
    got = np.empty_like(I)
    assert hip.hipMemcpy(got.ctypes.data_as(C.c_void_p), dO, nbytes, D2H) == 0
    want = np.stack([orcs[c].update(I[c], Q[c]).reshape(n_blk, 128) for c in range(n_ch)])
    assert np.array_equal(got, want)
    # timing is opt-in (no event packets around the kernels by default)
    assert batch.last_kernel_ms() < 0
    batch.set_launch_timing(True)
    batch.update_device(dI.value, dQ.value, dO.value, n_blk, stream.value)
    assert batch.last_kernel_ms() > 0
    batch.set_launch_timing(False)
    assert batch.last_kernel_ms() < 0
    # per-launch pairs
    batch.kernel_timing_begin(3)
    for _ in range(3):
        batch.update_device(dI.value, dQ.value, dO.value, n_blk, stream.value)
    ms = batch.kernel_timing_end(3)
    assert len(ms) == 3 and (ms > 0).all()
    # one pair around a region of calls (bench.py's kernel_ms): the four kernels' time.  (Compared loosely: a per-launch pair above carries
    # its own event packets -- 0.24-0.34 ms per call here against 0.19 inside a region -- so "4 x the fastest timed launch" is no lower bound;
    # the 0.8 factor this line had failed 4 runs of 14 on one box.)
    batch.region_timing_begin(stream.value)
    for _ in range(4):
        batch.update_device(dI.value, dQ.value, dO.value, n_blk, stream.value)
    total, calls = batch.region_timing_end()
    assert calls == 4 and total > 1.5 * float(ms.min()) and total < 4 * 4 * float(ms.max())
    for p_ in (dI, dQ, dO):
        hip.hipFree(p_)
    batch.close()


def test_full_size_c2_batch(gpu, ao):
    """BASELINE config 2 at full size (65,536 channels): a sample of channels against the oracle, plus
    size-independent properties over ALL channels: tiled duplicate channels give identical rows, and the
    batch equals a smaller batch of its first channels."""
    from audiosdr_amd.synth import make_iq
    n_ch, uniq, n_blk = 65536, 2048, 4
    I, Q = make_iq(uniq, n_blk, fc=6290.0, A=0.25, impulse_every=1000)
    I = np.tile(I, (n_ch // uniq, 1, 1)); Q = np.tile(Q, (n_ch // uniq, 1, 1))
    batch = gpu.AudioSDRBatch(n_ch)
    batch.setDemodMode(1); batch.enableAudioFilter()
    got = batch.update(I, Q)
    assert got.any()
    assert np.array_equal(got[:uniq], got[uniq:2 * uniq]) and np.array_equal(got[:uniq], got[-uniq:])
    for c in list(range(0, uniq, 97)) + [uniq - 1]:
        o = ao.OracleSDR(); o.setDemodMode(1); o.enableAudioFilter()
        assert np.array_equal(got[c].reshape(-1), o.update(I[c], Q[c])), c
    small = gpu.AudioSDRBatch(100)
    small.setDemodMode(1); small.enableAudioFilter()
    assert np.array_equal(small.update(I[:100], Q[:100]), got[:100])
    st = batch.read_status()
    assert st["agc_active"].all()
    small.close(); batch.close()


def test_sam_lock_fraction_c3_sample(gpu, ao):
    """BASELINE config 3 settings on a 4,096-channel sample with carriers offset by (c mod 7 - 3)*50 Hz: every PLL locks
    within 12 blocks; lock flags, PLL frequencies and audio equal the oracle for sampled channels."""
    from audiosdr_amd.synth import make_iq
    n_ch, n_blk = 4096, 12
    fc = 6890.0 + (np.arange(n_ch) % 7 - 3) * 50.0
    I, Q = make_iq(n_ch, n_blk, fc=fc, A=0.3, m=0.5, fm=400.0)
    batch = gpu.AudioSDRBatch(n_ch)
    batch.setDemodMode(5); batch.setNoiseBlankerThresholdDb(10.0); batch.enableAudioFilter(); batch.setAudioFilter(0)
    got = batch.update(I, Q)
    st = batch.read_status()
    assert st["sam_locked"].mean() == 1.0
    assert np.all(np.abs(st["sam_frequency"] - fc) < 30.0)
    for c in range(0, n_ch, 311):
        o = ao.OracleSDR(); o.setDemodMode(5); o.setNoiseBlankerThresholdDb(10.0); o.enableAudioFilter(); o.setAudioFilter(0)
        assert np.array_equal(got[c].reshape(-1), o.update(I[c], Q[c]))
        assert f32_bits(st["sam_frequency"][c]) == f32_bits(np.float32(o.getSAMfrequency()))
    batch.close()


def test_comparison_paths_still_bit_exact(gpu):
    """Two code paths are kept behind environment switches for A/B measurements (read once per process, hence the child process):
    ASDR_SAM_FUSED=1 = the fused 4-wave SAM kernel instead of the pre | PLL | post launches, ASDR_NO_STREAM_PIPELINE=1 = the in-kernel
    block loop instead of the block pipeline.  Both must stay bit-exact against the oracle."""
    import os
    import subprocess
    import sys
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import audiosdr_amd as A
from oracle import asdr_oracle as ao
from audiosdr_amd.synth import make_iq
n_ch, T = 64, 12
fc = 6890.0 + (np.arange(n_ch) %% 5 - 2) * 40.0
I, Q = make_iq(n_ch, T, fc=fc, A=0.3, m=0.4, noise=0.02)
for mode in (5, 1):                                   # SAM (fused kernel), LSB (in-kernel block loop, 8 uniform waves x 12 blocks)
    b = A.AudioSDRBatch(n_ch); b.setDemodMode(mode); b.enableAudioFilter()
    got = b.update(I, Q)
    assert b.stream_pipeline_launches() == 0
    for c in range(n_ch):
        o = ao.OracleSDR(); o.setDemodMode(mode); o.enableAudioFilter()
        assert np.array_equal(got[c], o.update(I[c], Q[c]).reshape(T, 128)), (mode, c)
    b.close()
print("ok")
''' % (os.path.join(os.path.dirname(__file__), ".."), os.path.dirname(__file__))
    env = dict(os.environ, ASDR_SAM_FUSED="1", ASDR_NO_STREAM_PIPELINE="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr
    # ... and the other way round: the pre | PLL | post launches for a batch below the size at which they are chosen (512 SAM channels)
    env = dict(os.environ, ASDR_SAM_SPLIT_MIN="1", ASDR_NO_STREAM_PIPELINE="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr


def test_als_kernel_kinds(gpu, ao):
    """The ALS filter runs in one of three kernel kinds (asdr_host.cpp kernel_kind): short tap sets (taps <= 64, delay + taps <= 65)
    on the compact 388-float rows, SAM channels with such a filter through the pre | PLL | post launches (>= 512 SAM channels in
    the batch), everything else on the 516-float rows.  One batch with all of them, filter lengths at the limits of the compact
    layout, notch / peak, adaptive / static, a 3-block call (one launch per block with the SAM launches) -- every variety vs the oracle,
    and the tap rows in HBM keep their natural order (a parameter change moves a channel between kinds mid-stream)."""
    from audiosdr_amd.synth import make_iq
    n_ch, T = 1400, 3
    I, Q = make_iq(n_ch, 3 * T, fc=6890.0 + (np.arange(n_ch) % 9 - 4) * 35.0, A=0.3, m=0.4, f2=7600.0, a2=0.12, noise=0.01)
    variety = [  # (mode, (M, lambda, delay) or None, peak, static)
        (5, None, False, False), (5, (64, 0.5, 1), False, False), (5, (65, 0.5, 0), False, False), (5, (100, 0.25, 7), True, False),
        (1, None, False, False), (1, (64, 0.5, 1), True, False), (1, (61, 0.3, 4), False, True), (1, (62, 0.5, 4), False, False),
        (4, (8, 0.5, 0), False, False), (0, (7, 0.5, 58), False, False), (6, (1, 0.5, 64), False, False), (2, (0, 0.5, 3), False, False),
        (5, (55, 0.5, 3), False, True), (3, (128, 0.05, 1), False, False),
    ]
    def configure(s, v, ch=None):
        kw = {} if ch is None else {"ch": ch}
        mode, par, peak, static = v
        s.setDemodMode(mode, **kw); s.setNoiseBlankerThresholdDb(10.0, **kw); s.enableALSfilter(**kw)
        if par is not None: s.setALSfilterParams(*par, **kw)
        if peak: s.setALSfilterPeak(**kw)
        if static: s.setALSfilterStatic(**kw)
    b = gpu.AudioSDRBatch(n_ch)
    b.setDemodMode(5); b.setNoiseBlankerThresholdDb(10.0); b.enableALSfilter()      # 1400 SAM + ALS channels: the three-launch path
    kind_of = {}
    for c in range(0, n_ch, 3):                                                     # every third channel: one of the varieties
        kind_of[c] = variety[(c // 3) % len(variety)]
        configure(b, kind_of[c], ch=c)
    got = [b.update(I[:, :T], Q[:, :T])]
    # mid-stream: a compact-layout channel gets a long filter (moves to the 516-float rows with its taps), and back
    b.setALSfilterParams(90, 0.5, 3, ch=1); got.append(b.update(I[:, T:2 * T], Q[:, T:2 * T]))
    b.setALSfilterParams(40, 0.5, 3, ch=1); got.append(b.update(I[:, 2 * T:], Q[:, 2 * T:]))
    got = np.concatenate(got, axis=1)
    checked = set()
    for c in list(range(0, 3 * len(variety) * 2, 3)) + [1, 2, n_ch - 1, n_ch - 2]:
        o = ao.OracleSDR()
        if c in kind_of: configure(o, kind_of[c])
        else: o.setDemodMode(5); o.setNoiseBlankerThresholdDb(10.0); o.enableALSfilter()
        if c == 1:
            w = [o.update(I[c, :T], Q[c, :T])]; o.setALSfilterParams(90, 0.5, 3)
            w.append(o.update(I[c, T:2 * T], Q[c, T:2 * T])); o.setALSfilterParams(40, 0.5, 3); w.append(o.update(I[c, 2 * T:], Q[c, 2 * T:]))
            want = np.concatenate(w)
        else:
            want = o.update(I[c], Q[c])
        assert np.array_equal(got[c].reshape(-1), want.reshape(-1)), (c, kind_of.get(c))
        checked.add(kind_of.get(c))
    assert len(checked) == len(variety) + 1
    b.close()


def test_als_as_a_launch_of_its_own(gpu, ao):
    """Channels with a short ALS filter (not SAM) can run as two launches: the chain up to the AGC as the plain kernel, then the filter
    + output stage on small LDS rows whose output overlays the consumed history (asdr_als_kernel; asdr_set_als_launch_form).  Forced
    here for a small batch: USB / LSB / CW / AM / WSPR groups of 8 with filter lengths and delays up to the compact layout's limits
    (M + delay = 65: the deepest reach into the overlaid history), notch / peak, adaptive / static taps, a muted group, an unknown mode
    (the kept audio row is the post-ALS one); single-block and 3-block calls, a filter parameter change and a disable / enable
    (taps + history zeroed, AudioSDR.cpp:384-391) mid-stream.  Every block of every channel, and the ALS-stage tap is NOT available
    in this form (stage taps keep the fused kernel), so int16 audio + status."""
    from audiosdr_amd.synth import make_iq
    n_ch, plan = 96, (1, 1, 3, 1, 2, 1, 3)
    total = sum(plan)
    I, Q = make_iq(n_ch, total, fc=6890.0 + (np.arange(n_ch) % 9 - 4) * 35.0, A=0.3, m=0.4, f2=7600.0, a2=0.12, noise=0.01, impulse_every=500)
    grp = lambda g: (lambda c: c // 8 == g)
    variety = [(1, None, False, False), (0, (64, 0.5, 1), False, False), (3, (1, 0.5, 64), True, False), (4, (55, 0.5, 3), False, False),
               (6, (60, 0.25, 5), False, True), (1, (8, 0.5, 57), True, True), (2, (0, 0.5, 3), False, False), (1, (33, 0.7, 32), False, False),
               (4, (64, 0.1, 0), True, False), (1, None, False, False), (0, (17, 0.5, 40), False, False), (1, (55, 0.5, 3), False, False)]
    setters = [S("setNoiseBlankerThresholdDb", 10.0), S("enableALSfilter")]
    for g, (mode, par, peak, static) in enumerate(variety):
        setters.append(S("setDemodMode", mode, sel=grp(g)))
        if par is not None: setters.append(S("setALSfilterParams", *par, sel=grp(g)))
        if peak: setters.append(S("setALSfilterPeak", sel=grp(g)))
        if static: setters.append(S("setALSfilterStatic", sel=grp(g)))
    setters += [S("enableAudioFilter", sel=grp(3)), S("setMute", 1, sel=grp(9))]
    batch, orcs = _mk(gpu, ao, n_ch, setters)
    batch.set_als_launch_form(8)
    script = {2: [S("setALSfilterParams", 40, 0.5, 20.0, sel=grp(0))], 3: [S("setDemodMode", 7, sel=grp(11))],
              4: [S("disableALSfilter", sel=grp(7))], 5: [S("enableALSfilter", sel=grp(7)), S("setDemodMode", 1, sel=grp(11))]}
    pos = 0
    for k, T in enumerate(plan):
        if k in script:
            apply_setters(batch, orcs, script[k])
        got = batch.update(I[:, pos:pos + T], Q[:, pos:pos + T])
        assert batch.schedule_layout()["als_two_launches"]
        for c in range(n_ch):
            want = orcs[c].update(I[c, pos:pos + T], Q[c, pos:pos + T]).reshape(T, 128)
            assert np.array_equal(got[c], want), "call %d (blocks %d..%d) ch %d (group %d)" % (k, pos, pos + T - 1, c, c // 8)
        pos += T
    compare_status(gpu, batch, orcs)
    batch.close()


def test_short_division_of_the_unit_gain_envelope_is_exhaustively_exact(gpu, tmp_path):
    """The blanker envelope of unit-gain waves (asdr_kernels.hip fast_sqrt1_short) replaces the IEEE division inside the reference's
    fast_sqrt_f32 (AudioSDR.h:434-446) by v_rcp_f32 + a Newton step + a residual correction.  v_rcp_f32 is a hardware approximation, so the
    proof is an exhaustive run ON the GPU: tools/ubench/sqrt_div_check.hip compares the RESULT for all 2^31 non-negative finite floats.
    They must agree for x = 0 and for every x >= 1e-30 (unit-gain samples give x = 0 or x >= 9.3e-10)."""
    import re
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools", "ubench", "sqrt_div_check.hip")
    exe = str(tmp_path / "sqrt_div_check")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-ffp-contract=off", src, "-o", exe], check=True, capture_output=True, timeout=300)
    out = subprocess.run([exe], check=True, capture_output=True, text=True, timeout=300).stdout
    m = re.search(r"mismatching inputs: (\d+) of \d+; \|x\| range of mismatches: \[(\S+) \(0x([0-9a-f]+)\), (\S+) \(0x([0-9a-f]+)\)\]", out)
    assert m, out
    n_bad, lo_bits, hi_bits = int(m.group(1)), int(m.group(3), 16), int(m.group(5), 16)
    if n_bad:
        assert lo_bits > 0, "x = 0 differs: %s" % out
        assert np.uint32(hi_bits).view(np.float32) < 1e-30, "a mismatch at or above 1e-30: %s" % out


@pytest.mark.parametrize("name", sorted(CASES))
def test_case_against_the_default_oracle_and_no_pll_stall(gpu, ao, name):
    """Every GPU test above compares with an oracle that models the product's ONE defined difference in the signal path -- the SAM PLL's
    bounded phase wrap (oracle/asdr_oracle.py PRODUCT_PLL_BOUND).  This test takes that switch away: every case of
    tests/cases.py in one multi-block call against the DEFAULT oracle (the reference's own unbounded loops, AudioSDR.cpp:735-736),
    bit for bit, and the oracle reports that none of its wrap loops ever reached the state in which the reference would hang
    (ao_pll_stalled() == 0) -- i.e. the defined difference is inert on everything the suite feeds the chain."""
    from audiosdr_amd.synth import make_iq
    n_ch, n_blk, setters, sig = CASES[name]
    I, Q = make_iq(n_ch, n_blk, **sig)
    batch = gpu.AudioSDRBatch(n_ch)
    orcs = [ao.OracleSDR(pll_wrap_bound=False) for _ in range(n_ch)]
    apply_setters(batch, orcs, setters)
    got = batch.update(I, Q)
    for c in range(n_ch):
        want = orcs[c].update(I[c], Q[c]).reshape(n_blk, 128)
        assert np.array_equal(got[c], want), "%s ch %d" % (name, c)
        assert orcs[c].pll_stalled() == 0, "%s ch %d: a reference wrap loop would not have ended" % (name, c)
    compare_status(gpu, batch, orcs)
    batch.close()
