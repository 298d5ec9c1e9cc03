"""SURVEY.md 8(f) row 1 on the GPU: strided streaming out of / into longer HBM rows and the capture sink
(one contiguous audio row per channel), bit-for-bit against the CPU oracle run block by block."""
import numpy as np
import pytest

from helpers import Hip, S, apply_setters

pytestmark = pytest.mark.gpu
WSPR = [S("enableAGC"), S("setAGCmode", 2), S("disableALSfilter"), S("disableNoiseBlanker"), S("setNoiseBlankerThresholdDb", 10.0),
        S("setInputGain", 1.0), S("setOutputGain", 0.5), S("setIQgainBalance", 1.020), S("setAudioFilter", 2),
        S("setDemodMode", 6), S("setMute", 0)]     # EXTRAS/BareBonesWSPR/BareBonesWSPR.ino:87-102,129


def _mk(gpu, ao, n_ch, setters):
    batch = gpu.AudioSDRBatch(n_ch)
    orcs = [ao.OracleSDR() for _ in range(n_ch)]
    apply_setters(batch, orcs, setters)
    return batch, orcs


def test_strided_rows_in_and_out(gpu, ao):
    """Blocks 3..6 of 10-block input rows -> blocks 2..5 of 7-block output rows; everything else untouched."""
    from audiosdr_amd.synth import make_iq
    n_ch, in_blk, out_blk, T = 19, 10, 7, 4
    I, Q = make_iq(n_ch, in_blk, fc=6290.0, A=0.25, noise=0.02, impulse_every=300)
    batch, orcs = _mk(gpu, ao, n_ch, [S("setDemodMode", 1), S("enableAudioFilter")])
    hip = Hip()
    dI, dQ = hip.upload(I), hip.upload(Q)
    dO = hip.malloc(n_ch * out_blk * 256)
    hip.fill(dO, 0x5A, n_ch * out_blk * 256)
    batch.update_device_strided(dI + 3 * 256, dQ + 3 * 256, dO + 2 * 256, T, in_blk, out_blk)
    batch.synchronize()
    got = hip.download(dO, (n_ch, out_blk, 128), np.int16)
    for c in range(n_ch):
        want = orcs[c].update(I[c, 3:3 + T], Q[c, 3:3 + T]).reshape(T, 128)
        assert np.array_equal(got[c, 2:2 + T], want), "ch %d" % c
    assert (got[:, :2] == 0x5A5A).all() and (got[:, 2 + T:] == 0x5A5A).all()
    with pytest.raises(gpu.AsdrError, match="stride"):
        batch.update_device_strided(dI, dQ, dO, 4, 3, 7)
    with pytest.raises(gpu.AsdrError, match="aligned"):
        batch.update_device_strided(dI + 2, dQ, dO, 1, in_blk, out_blk)
    hip.free_all(); batch.close()


def test_capture_sink_ragged_launches(gpu, ao):
    """A WSPR-configured batch streamed into the capture sink in launches of 1, 5, 17 and 9 blocks: every
    channel's row equals the oracle's block-by-block audio; position/overflow/rewind behave."""
    from audiosdr_amd.synth import make_iq
    n_ch, total = 21, 32
    I, Q = make_iq(n_ch, total, fc=6890.0 - 1500.0, A=0.02, noise=0.05)
    batch, orcs = _mk(gpu, ao, n_ch, WSPR)
    hip = Hip()
    with pytest.raises(gpu.AsdrError, match="not open"):
        batch.capture_update_device(0, 0, 1)
    batch.capture_open(40)
    assert batch.capture_capacity == 40 and batch.capture_position == 0
    pos = 0
    for T in (1, 5, 17, 9):
        dI, dQ = hip.upload(I[:, pos:pos + T]), hip.upload(Q[:, pos:pos + T])
        batch.capture_update_device(dI, dQ, T)
        pos += T
        assert batch.capture_position == pos
    for c in range(n_ch):
        want = orcs[c].update(I[c], Q[c])
        assert np.array_equal(batch.capture_read(c), want), "ch %d" % c
        assert np.array_equal(batch.capture_read(c, 6, 3), want[6 * 128:9 * 128])
    # the sink is plain HBM: a decoder can read rows in place
    whole = hip.download(batch.capture_device_ptr(), (n_ch, 40, 128), np.int16)
    assert np.array_equal(whole[3, :total].reshape(-1), batch.capture_read(3))
    # missing input appends nothing; overflow is refused without processing
    batch.capture_update_device(0, dQ, 1)
    assert batch.capture_position == total
    dI, dQ = hip.upload(I[:, :9]), hip.upload(Q[:, :9])
    with pytest.raises(gpu.AsdrError, match="overflow"):
        batch.capture_update_device(dI, dQ, 9)
    assert batch.capture_position == total
    with pytest.raises(gpu.AsdrError, match="beyond"):
        batch.capture_read(0, 30, 5)
    # rewind keeps channel state: the next block continues the stream
    batch.capture_rewind()
    batch.capture_update_device(dI, dQ, 9)
    for c in (0, n_ch - 1):
        assert np.array_equal(batch.capture_read(c), orcs[c].update(I[c, :9], Q[c, :9]))
    batch.capture_close()
    assert batch.capture_device_ptr() == 0
    hip.free_all(); batch.close()


def test_capture_long_stream_from_resident_input(gpu, ao):
    """512 blocks per channel streamed 64 at a time out of ONE resident input buffer (in_stride = 512) into the sink."""
    from audiosdr_amd.synth import make_iq
    n_ch, total, T = 16, 512, 64
    I, Q = make_iq(n_ch, total, fc=6890.0 - 1500.0 + 40.0, A=0.05, noise=0.05)
    batch, orcs = _mk(gpu, ao, n_ch, WSPR)
    hip = Hip()
    dI, dQ = hip.upload(I), hip.upload(Q)
    batch.capture_open(total)
    for pos in range(0, total, T):
        batch.capture_update_device(dI + pos * 256, dQ + pos * 256, T, in_stride_blocks=total)
    for c in (0, 7, 15):
        assert np.array_equal(batch.capture_read(c), orcs[c].update(I[c], Q[c])), "ch %d" % c
    hip.free_all(); batch.close()


# ---- block pipeline (asdr_stream_kernel): a multi-block call on a small batch of uniform SSB-class waves ------------------
def _compare_status(batch, orcs):
    st = batch.read_status()
    for c, o in enumerate(orcs):
        assert int(st["agc_active"][c]) == o.AGCisActive(), "AGCisActive ch %d" % c
        assert int(st["nb_detected"][c]) == o.NoiseBlankerDetection(), "NoiseBlankerDetection ch %d" % c


def test_block_pipeline_parity_and_hand_over(gpu, ao):
    """64 USB channels (blanker with impulses, audio filter, AGC): calls of 40, 1, 24, 3 and 16 blocks.  The 40-, 24- and 16-block
    calls run as the three-role block pipeline, the others block by block: every block of every channel equals the oracle, i.e.
    the per-channel state is handed over correctly in both directions, and the status bits (written by two different roles) match."""
    from audiosdr_amd.synth import make_iq
    n_ch, plan = 64, (40, 1, 24, 3, 16)
    total = sum(plan)
    I, Q = make_iq(n_ch, total, fc=6290.0, A=0.25, noise=0.02, impulse_every=777)
    batch, orcs = _mk(gpu, ao, n_ch, [S("setDemodMode", 1), S("enableAudioFilter"), S("setNoiseBlankerThresholdDb", 10.0)])
    hip = Hip()
    pos = 0
    for T in plan:
        dI, dQ = hip.upload(I[:, pos:pos + T]), hip.upload(Q[:, pos:pos + T])
        dO = hip.malloc(n_ch * T * 256)
        batch.update_device(dI, dQ, dO, T)
        batch.synchronize()
        got = hip.download(dO, (n_ch, T, 128), np.int16)
        for c in range(n_ch):
            want = orcs[c].update(I[c, pos:pos + T], Q[c, pos:pos + T]).reshape(T, 128)
            assert np.array_equal(got[c], want), "call of %d blocks at %d, ch %d" % (T, pos, c)
        pos += T
    assert batch.stream_pipeline_launches() == 3
    _compare_status(batch, orcs)
    hip.free_all(); batch.close()


def test_block_pipeline_key_groups(gpu, ao):
    """Four key groups of 16 channels (USB / LSB with the blanker off / CW with the audio filter / WSPR as the reference's WSPR
    receiver configures it), each two uniform waves: one sub-range of uniform waves with different modes and enables per wave."""
    from audiosdr_amd.synth import make_iq
    n_ch, T = 64, 33
    I, Q = make_iq(n_ch, T, fc=6290.0, A=0.2, noise=0.03, impulse_every=500, m=0.3)
    grp = lambda g: (lambda c: c // 16 == g)
    setters = [S("setDemodMode", 0, sel=grp(0)), S("setNoiseBlankerThresholdDb", 9.0, sel=grp(0)),
               S("setDemodMode", 1, sel=grp(1)), S("disableNoiseBlanker", sel=grp(1)), S("setAGCmode", 1, sel=grp(1)),
               S("setDemodMode", 2, sel=grp(2)), S("enableAudioFilter", sel=grp(2)), S("setAudioFilter", 0, sel=grp(2)),
               S("setOutputGain", 0.7, sel=grp(2))]
    batch, orcs = _mk(gpu, ao, n_ch, setters)
    apply_setters(batch, orcs, [(m, a, (lambda c, s=s: c // 16 == 3 and (s is None or s(c)))) for (m, a, s) in WSPR])
    hip = Hip()
    dI, dQ = hip.upload(I), hip.upload(Q)
    dO = hip.malloc(n_ch * T * 256)
    batch.update_device(dI, dQ, dO, T)
    batch.synchronize()
    got = hip.download(dO, (n_ch, T, 128), np.int16)
    assert batch.stream_pipeline_launches() == 1
    for c in range(n_ch):
        want = orcs[c].update(I[c], Q[c]).reshape(T, 128)
        assert np.array_equal(got[c], want), "ch %d" % c
    _compare_status(batch, orcs)
    hip.free_all(); batch.close()


def test_block_pipeline_is_not_used_where_it_does_not_apply(gpu, ao):
    """A SAM channel group (its PLL is a 128-step chain per block: no role set), an ALS filter, stage taps or a short call keep the
    block-by-block path (same results either way).  A ragged channel count no longer does (round 4): the 7 whole waves of 61 USB
    channels take the pipeline, the 5 left over run the call on the in-kernel block loop beside it."""
    from audiosdr_amd.synth import make_iq
    T = 12
    for n_ch, setters, taps, expect in ((64, [S("setDemodMode", 5)], False, 0), (64, [S("setDemodMode", 4), S("enableALSfilter")], False, 0),
                                        (64, [S("setDemodMode", 1)], True, 0), (61, [S("setDemodMode", 1)], False, 1)):
        I, Q = make_iq(n_ch, T + 4, fc=6290.0, A=0.25)
        batch, orcs = _mk(gpu, ao, n_ch, setters)
        if taps:
            batch.enable_taps(True)
        hip = Hip()
        dI, dQ = hip.upload(I[:, :T]), hip.upload(Q[:, :T])
        dI2, dQ2 = hip.upload(I[:, T:]), hip.upload(Q[:, T:])
        dO, dO2 = hip.malloc(n_ch * T * 256), hip.malloc(n_ch * 4 * 256)
        batch.update_device(dI, dQ, dO, T)
        batch.update_device(dI2, dQ2, dO2, 4)       # short call: never the pipeline
        batch.synchronize()
        assert batch.stream_pipeline_launches() == expect
        got = np.concatenate([hip.download(dO, (n_ch, T, 128), np.int16), hip.download(dO2, (n_ch, 4, 128), np.int16)], axis=1)
        for c in range(0, n_ch, 7 if expect == 0 else 1):
            assert np.array_equal(got[c], orcs[c].update(I[c], Q[c]).reshape(T + 4, 128)), (n_ch, c)
        hip.free_all(); batch.close()


def test_block_pipeline_am_role_set(gpu, ao):
    """AM through the block pipeline (round 3): role 2 runs mixer + image filter + envelope detector + carrier tracker
    (AudioSDR.cpp:132-143), the block's carrier level crosses to role 3 beside the audio row, whose AGC takes twice that level in
    place of |x| (:407-409).  96 channels in three key groups -- AM with blanker + audio filter + AGC, AM with AGC mode 1 and no
    blanker, USB (so that SSB and AM waves share one pipeline launch; the oscillator role serves the first group) -- with
    different carrier levels and modulation depths; calls of 40, 1, 24, 2 and 16 blocks: the carrier tracker's state, the image
    filter's state and the AGC's state are handed over in both directions.  Every block of every channel and the getters
    (AM carrier level included) against the oracle."""
    from audiosdr_amd.synth import make_iq
    n_ch, plan = 96, (40, 1, 24, 2, 16)
    total = sum(plan)
    fc = 6890.0 + (np.arange(n_ch) % 5 - 2) * 30.0
    I, Q = make_iq(n_ch, total, fc=fc, A=0.05 + 0.5 * (np.arange(n_ch) % 7) / 7.0, m=0.6, fm=440.0, noise=0.01, impulse_every=900)
    grp = lambda g: (lambda c: c // 32 == g)
    setters = [S("setDemodMode", 4, sel=grp(0)), S("enableAudioFilter", sel=grp(0)), S("setNoiseBlankerThresholdDb", 10.0, sel=grp(0)),
               S("setDemodMode", 4, sel=grp(1)), S("disableNoiseBlanker", sel=grp(1)), S("setAGCmode", 1, sel=grp(1)), S("setAGChangTime", 5.0, sel=grp(1)),
               S("setDemodMode", 1, sel=grp(2)), S("enableAudioFilter", sel=grp(2))]
    batch, orcs = _mk(gpu, ao, n_ch, setters)
    hip = Hip()
    pos = 0
    for T in plan:
        dI, dQ = hip.upload(I[:, pos:pos + T]), hip.upload(Q[:, pos:pos + T])
        dO = hip.malloc(n_ch * T * 256)
        batch.update_device(dI, dQ, dO, T)
        batch.synchronize()
        got = hip.download(dO, (n_ch, T, 128), np.int16)
        for c in range(n_ch):
            want = orcs[c].update(I[c, pos:pos + T], Q[c, pos:pos + T]).reshape(T, 128)
            assert np.array_equal(got[c], want), "call of %d blocks at %d, ch %d" % (T, pos, c)
        pos += T
    assert batch.stream_pipeline_launches() == 3
    assert batch.stream_pipeline_recoveries() == 0
    from helpers import compare_status
    compare_status(gpu, batch, orcs)
    hip.free_all(); batch.close()


def test_block_pipeline_am_timeout_is_recovered(gpu, ao):
    """The transaction with an all-AM batch: injected timeouts (poll limit 30) are recovered in-stream; audio and getters exact."""
    from audiosdr_amd.synth import make_iq
    n_ch, plan = 64, (20, 18)
    total = sum(plan)
    I, Q = make_iq(n_ch, total, fc=6890.0, A=0.3, m=0.5, fm=300.0, noise=0.01)
    batch, orcs = _mk(gpu, ao, n_ch, [S("setDemodMode", 4), S("enableAudioFilter"), S("setNoiseBlankerThresholdDb", 10.0)])
    batch.debug_set_stream_spin_limit(30)
    hip = Hip()
    outs, pos = [], 0
    for T in plan:
        dI, dQ = hip.upload(I[:, pos:pos + T]), hip.upload(Q[:, pos:pos + T])
        dO = hip.malloc(n_ch * T * 256)
        batch.update_device(dI, dQ, dO, T)
        outs.append((dO, pos, T))
        pos += T
    batch.synchronize()
    assert batch.stream_pipeline_launches() == 2 and batch.stream_pipeline_recoveries() >= 1
    gots = [(hip.download(dO, (n_ch, T, 128), np.int16), p0, T) for dO, p0, T in outs]
    for c in range(n_ch):
        want = orcs[c].update(I[c], Q[c]).reshape(total, 128)
        for got, p0, T in gots:
            assert np.array_equal(got[c], want[p0:p0 + T]), "call at block %d, ch %d" % (p0, c)
    from helpers import compare_status
    compare_status(gpu, batch, orcs)
    hip.free_all(); batch.close()


@pytest.mark.parametrize("n_ch,T,pipelined", [(8, 9, True), (672, 11, True), (680, 9, True), (4096, 9, True), (4104, 8, False)])
def test_block_pipeline_sizes(gpu, ao, n_ch, T, pipelined):
    """One channel group (a single wave per role), 84 groups (one pipeline workgroup per compute unit), 85, the largest batch the
    pipeline takes on an MI355X (512 groups = 4,096 receivers: 3 x 512 workgroups, the occupancy query's limit) and one group more
    (in-kernel block loop).  A mode change between two calls resets filter state through the usual path."""
    from audiosdr_amd.synth import make_iq
    I, Q = make_iq(n_ch, 2 * T, fc=6290.0, A=0.25, noise=0.02)
    batch, orcs = _mk(gpu, ao, n_ch, [S("setDemodMode", 1), S("enableAudioFilter")])
    hip = Hip()
    sample = sorted(set([0, 1, 7, n_ch // 2, n_ch - 1]))
    for call in range(2):
        if call == 1:
            apply_setters(batch, orcs, [S("setDemodMode", 0)])        # USB -> LSB... (0 = USB): zeroes the IF state (.cpp:191-218)
        sl = slice(call * T, (call + 1) * T)
        dI, dQ = hip.upload(I[:, sl]), hip.upload(Q[:, sl])
        dO = hip.malloc(n_ch * T * 256)
        batch.update_device(dI, dQ, dO, T)
        batch.synchronize()
        got = hip.download(dO, (n_ch, T, 128), np.int16)
        for c in sample:
            want = orcs[c].update(I[c, sl], Q[c, sl]).reshape(T, 128)
            assert np.array_equal(got[c], want), "call %d ch %d" % (call, c)
    assert batch.stream_pipeline_launches() == (2 if pipelined else 0)
    assert batch.stream_pipeline_recoveries() == 0
    hip.free_all(); batch.close()


@pytest.mark.parametrize("mode,n_ch", [(0, 64), (1, 64), (-1, 64), (1, 1600), (-1, 1600)])
def test_block_pipeline_fir_helper_forms(gpu, ao, mode, n_ch):
    """The role-2 workgroup's Hilbert FIR with ONE helper wave (128-thread workgroups) and with THREE (asdr_stream_kernel_h3: the FIR in quarters),
    forced either way and by the rule (three while every pipeline workgroup has a compute unit to itself: 64 channels yes, 1,600 no -- not even
    forced: that form is resident one workgroup per compute unit): every block
    of the sampled channels equals the oracle, LSB / USB alternating between the groups so that both sideband signs pass through every helper."""
    from audiosdr_amd.synth import make_iq
    T = 12
    I, Q = make_iq(n_ch, T, fc=6290.0, A=0.25, noise=0.02, impulse_every=500)
    batch, orcs = _mk(gpu, ao, n_ch, [S("setDemodMode", 1), S("enableAudioFilter"), S("setNoiseBlankerThresholdDb", 10.0)])
    batch.set_stream_fir_helpers(mode)
    hip = Hip()
    dI, dQ = hip.upload(I), hip.upload(Q)
    dO = hip.malloc(n_ch * T * 256)
    batch.update_device(dI, dQ, dO, T)
    batch.synchronize()
    got = hip.download(dO, (n_ch, T, 128), np.int16)
    for c in sorted(set(list(range(0, 16)) + [n_ch // 2, n_ch - 9, n_ch - 1])):
        want = orcs[c].update(I[c], Q[c]).reshape(T, 128)
        assert np.array_equal(got[c], want), "ch %d" % c
    assert batch.stream_pipeline_launches() == 1 and batch.stream_pipeline_recoveries() == 0
    three = mode != 0 and n_ch == 64   # (1,600 channels = 600 pipeline workgroups: more than the three-helper form's one workgroup per compute unit, forced or not)
    assert batch.stream_pipeline_h3_calls() == (1 if three else 0)
    hip.free_all(); batch.close()


@pytest.mark.parametrize("limit", [1, 40])
def test_block_pipeline_timeout_is_recovered_in_stream(gpu, ao, limit):
    """The pipeline as a transaction (asdr.h, asdr_kernels.hip): with the poll limit of the bounded waits shrunk to `limit` the
    roles give up at once (1) or somewhere inside the call (40) -- as they would if the GPU were shared and the 3 w + 1 workgroups
    not co-resident.  The launches enqueued behind the pipeline restore the channels' state from the snapshot and run the call
    on the in-kernel block loop, all on the caller's stream and with no host synchronisation in between: three calls back to back
    (pipeline with injected timeouts, a one-block call, pipeline again), every block of every channel against the oracle, state
    (status getters) included; then the limit goes back to its default and the pipeline completes on its own."""
    from audiosdr_amd.synth import make_iq
    n_ch, plan = 64, (24, 1, 17, 20)
    total = sum(plan)
    I, Q = make_iq(n_ch, total, fc=6290.0, A=0.25, noise=0.02, impulse_every=777)
    batch, orcs = _mk(gpu, ao, n_ch, [S("setDemodMode", 1), S("enableAudioFilter"), S("setNoiseBlankerThresholdDb", 10.0)])
    assert batch.stream_pipeline_max_groups() >= 8
    batch.debug_set_stream_spin_limit(limit)
    hip = Hip()
    outs, pos = [], 0
    for k, T in enumerate(plan):
        if k == 3:
            batch.synchronize()
            assert batch.stream_pipeline_recoveries() >= 1          # the injected timeouts were seen and recovered ...
            rec = batch.stream_pipeline_recoveries()
            batch.debug_set_stream_spin_limit(0)                    # ... and with the default limit the pipeline runs through
        dI, dQ = hip.upload(I[:, pos:pos + T]), hip.upload(Q[:, pos:pos + T])
        dO = hip.malloc(n_ch * T * 256)
        hip.fill(dO, 0x11, n_ch * T * 256)
        batch.update_device(dI, dQ, dO, T)                          # no synchronisation between the calls
        outs.append((dO, pos, T))
        pos += T
    batch.synchronize()
    assert batch.stream_pipeline_recoveries() == rec
    assert batch.stream_pipeline_launches() == 3
    for dO, p0, T in outs:
        got = hip.download(dO, (n_ch, T, 128), np.int16)
        for c in range(n_ch):
            if p0 == 0:
                want = orcs[c].update(I[c, :total], Q[c, :total]).reshape(total, 128)
                orcs[c]._want = want
            assert np.array_equal(got[c], orcs[c]._want[p0:p0 + T]), "call at block %d (%d blocks), ch %d" % (p0, T, c)
    _compare_status(batch, orcs)
    hip.free_all(); batch.close()


@pytest.mark.parametrize("limit", [0, 1])
def test_block_pipeline_with_odd_channels_beside_it(gpu, ao, limit):
    """A small batch whose schedule is NOT one sub-range still takes the pipeline for its uniform SSB / AM waves: 200 USB receivers
    with an AM one, a SAM one, one with the ALS filter and one with other AGC settings among them -- the odd channels (a SAM and an
    ALS wave's worth of general-kernel remainders) run the same call on the in-kernel block loop beside the pipeline.  Every channel,
    every block and the status words against the oracle; with the poll limit at 1 the pipeline part is recovered in stream while the
    side launches are untouched."""
    from audiosdr_amd.synth import make_iq
    n_ch, plan = 204, (20, 1, 9)
    total = sum(plan)
    fc = 6290.0 + 15.0 * (np.arange(n_ch) % 3)
    I, Q = make_iq(n_ch, total, fc=fc, A=0.25, m=0.3, noise=0.02, impulse_every=1300)
    setters = [S("setDemodMode", 1), S("enableAudioFilter"), S("setNoiseBlankerThresholdDb", 10.0),
               S("setDemodMode", 4, sel=lambda c: c == 77), S("setDemodMode", 5, sel=lambda c: c == 130),
               S("enableALSfilter", sel=lambda c: c == 5), S("setAGCthreshold", -40.0, sel=lambda c: c == 190)]
    batch, orcs = _mk(gpu, ao, n_ch, setters)
    batch.debug_set_stream_spin_limit(limit)
    hip = Hip()
    outs, pos = [], 0
    for T in plan:
        dI, dQ = hip.upload(I[:, pos:pos + T]), hip.upload(Q[:, pos:pos + T])
        dO = hip.malloc(n_ch * T * 256)
        batch.update_device(dI, dQ, dO, T)
        outs.append((dO, pos, T)); pos += T
    batch.synchronize()
    assert batch.stream_pipeline_launches() == 2                     # the 20- and the 9-block call
    assert (batch.stream_pipeline_recoveries() >= 1) == (limit == 1)
    lay = batch.schedule_layout()
    assert lay["plain"] >= 192 and lay["remainders"] > 0
    want = [orcs[c].update(I[c], Q[c]).reshape(total, 128) for c in range(n_ch)]
    for dO, p0, T in outs:
        got = hip.download(dO, (n_ch, T, 128), np.int16)
        for c in range(n_ch):
            assert np.array_equal(got[c], want[c][p0:p0 + T]), "call at block %d, ch %d" % (p0, c)
    _compare_status(batch, orcs)
    hip.free_all(); batch.close()


@pytest.mark.parametrize("als", [False, True])
def test_sam_role_streams(gpu, ao, als):
    """SAM role streams (include/asdr.h asdr_sam_role_calls): in a multi-block call of a bank of SAM channels configured alike
    (>= 512: the three-launch form) the pre | PLL | post roles of consecutive blocks overlap on three streams.  What could go wrong is
    exactly what the test drives: the lock flag changes from block to block (the carrier jumps between an offset the PLL holds and one
    it cannot, at splice points that differ from channel to channel), so post(k) must see PLL(k)'s flag while PLL(k + 1) already runs;
    the tiles alternate between two sets; the status word is updated by three roles.  Calls of 12, 1, 7, 2 and 9 blocks: role
    streams and the ordinary one-call-one-block form hand the state to each other.  Every block of every channel, and the getters."""
    from audiosdr_amd.synth import make_iq
    n_ch, plan = 512, (12, 1, 7, 2, 9)
    total = sum(plan)
    near = 6890.0 + (np.arange(n_ch) % 7 - 3) * 40.0
    Ia, Qa = make_iq(n_ch, total, fc=near, A=0.3, m=0.5, fm=400.0, noise=0.005)
    Ib, Qb = make_iq(n_ch, total, fc=near + 2500.0, A=0.25, m=0.3, fm=300.0, noise=0.005)   # 2.5 kHz off: no lock
    I, Q = Ia.copy(), Qa.copy()
    for c in range(n_ch):
        for b in range(total):
            if ((b + c % 5) // 3) % 3 == 2:       # every channel loses its carrier for three blocks out of nine, shifted by channel
                I[c, b], Q[c, b] = Ib[c, b], Qb[c, b]
    setters = [S("setDemodMode", 5), S("setNoiseBlankerThresholdDb", 10.0), S("enableAudioFilter"), S("setAudioFilter", 0)]
    if als:
        setters.append(S("enableALSfilter"))
    batch, orcs = _mk(gpu, ao, n_ch, setters)
    hip = Hip()
    pos, locks = 0, set()
    for T in plan:
        dI, dQ = hip.upload(I[:, pos:pos + T]), hip.upload(Q[:, pos:pos + T])
        dO = hip.malloc(n_ch * T * 256)
        batch.update_device(dI, dQ, dO, T)
        batch.synchronize()
        got = hip.download(dO, (n_ch, T, 128), np.int16)
        for c in range(n_ch):
            want = orcs[c].update(I[c, pos:pos + T], Q[c, pos:pos + T]).reshape(T, 128)
            assert np.array_equal(got[c], want), "call of %d blocks at %d, ch %d" % (T, pos, c)
        st = batch.read_status()
        for c in range(0, n_ch, 7):
            assert int(st["sam_locked"][c]) == orcs[c].getSAMphaseLockStatus(), "lock flag after block %d, ch %d" % (pos + T - 1, c)
            locks.add(int(st["sam_locked"][c]))
        pos += T
    assert locks == {0, 1}                                  # both lock states were seen at call boundaries
    assert batch.sam_role_calls() == sum(1 for T in plan if T >= 2)
    from helpers import compare_status
    compare_status(gpu, batch, orcs)
    hip.free_all(); batch.close()


def test_sam_role_streams_in_chunks(gpu, ao):
    """... and in CHUNKS (include/asdr.h asdr_sam_chunk_calls): calls of 16 blocks or more of a uniform SAM bank run pre | PLL | post launches
    of 8 blocks each, their block loops kept, through 32 tile sets.  Same drive as above -- the lock flag changes from block to block at
    channel-dependent splice points, so post(k) must read PLL(k)'s flag and tile of the RIGHT set while later chunks are in flight -- with calls
    of 19, 1, 43 (more than the sets: the pre role waits for the post role) and 16 blocks handing the state to each other and to the
    one-call-one-block form."""
    from audiosdr_amd.synth import make_iq
    n_ch, plan = 512, (19, 1, 43, 16)
    total = sum(plan)
    near = 6890.0 + (np.arange(n_ch) % 7 - 3) * 40.0
    Ia, Qa = make_iq(n_ch, total, fc=near, A=0.3, m=0.5, fm=400.0, noise=0.005)
    Ib, Qb = make_iq(n_ch, total, fc=near + 2500.0, A=0.25, m=0.3, fm=300.0, noise=0.005)   # 2.5 kHz off: no lock
    I, Q = Ia.copy(), Qa.copy()
    for c in range(n_ch):
        for b in range(total):
            if ((b + c % 5) // 3) % 3 == 2:
                I[c, b], Q[c, b] = Ib[c, b], Qb[c, b]
    setters = [S("setDemodMode", 5), S("setNoiseBlankerThresholdDb", 10.0), S("enableAudioFilter"), S("setAudioFilter", 0)]
    batch, orcs = _mk(gpu, ao, n_ch, setters)
    hip = Hip()
    pos, locks = 0, set()
    for T in plan:
        dI, dQ = hip.upload(I[:, pos:pos + T]), hip.upload(Q[:, pos:pos + T])
        dO = hip.malloc(n_ch * T * 256)
        batch.update_device(dI, dQ, dO, T)
        batch.synchronize()
        got = hip.download(dO, (n_ch, T, 128), np.int16)
        for c in range(0, n_ch, 3):
            want = orcs[c].update(I[c, pos:pos + T], Q[c, pos:pos + T]).reshape(T, 128)
            assert np.array_equal(got[c], want), "call of %d blocks at %d, ch %d: first differing block %d" % (T, pos, c, int(np.nonzero((got[c] != want).any(axis=1))[0][0]))
        st = batch.read_status()
        for c in range(0, n_ch, 21):
            assert int(st["sam_locked"][c]) == orcs[c].getSAMphaseLockStatus(), "lock flag after block %d, ch %d" % (pos + T - 1, c)
            locks.add(int(st["sam_locked"][c]))
        pos += T
    assert batch.sam_chunk_calls() == sum(1 for T in plan if T >= 16)
    assert batch.sam_role_calls() == sum(1 for T in plan if T >= 2)
    hip.free_all(); batch.close()


def test_sam_role_streams_chunk_tail_of_one_block(gpu, ao):
    """Calls whose LAST chunk holds one block (17 = 2 x 8 + 1, 25, 41 blocks): round 5 launched the one-block pre / post kernels for that chunk,
    which know nothing of tile sets -- the pre role wrote set 0 while the PLL kernel read set (first block of the chunk) % 32, so the call's last
    block and the PLL state behind it were wrong unless that set happened to be 0.  The looped kernels run every chunk of
    a chunked call now, whatever its length; the calls hand the state to each other, so a wrong PLL state would also show in the next call."""
    from audiosdr_amd.synth import make_iq
    n_ch, plan = 512, (17, 25, 41, 2)   # (512 channels: the smallest bank that runs SAM as three launches)
    total = sum(plan)
    near = 6890.0 + (np.arange(n_ch) % 7 - 3) * 40.0
    Ia, Qa = make_iq(n_ch, total, fc=near, A=0.3, m=0.5, fm=400.0, noise=0.005)
    Ib, Qb = make_iq(n_ch, total, fc=near + 2500.0, A=0.25, m=0.3, fm=300.0, noise=0.005)   # 2.5 kHz off: no lock
    I, Q = Ia.copy(), Qa.copy()
    for c in range(n_ch):
        for b in range(total):
            if ((b + c % 5) // 3) % 3 == 2:
                I[c, b], Q[c, b] = Ib[c, b], Qb[c, b]
    setters = [S("setDemodMode", 5), S("setNoiseBlankerThresholdDb", 10.0), S("enableAudioFilter"), S("setAudioFilter", 0)]
    batch, orcs = _mk(gpu, ao, n_ch, setters)
    hip = Hip()
    pos = 0
    for T in plan:
        dI, dQ = hip.upload(I[:, pos:pos + T]), hip.upload(Q[:, pos:pos + T])
        dO = hip.malloc(n_ch * T * 256)
        batch.update_device(dI, dQ, dO, T)
        batch.synchronize()
        got = hip.download(dO, (n_ch, T, 128), np.int16)
        for c in range(0, n_ch, 3):
            want = orcs[c].update(I[c, pos:pos + T], Q[c, pos:pos + T]).reshape(T, 128)
            assert np.array_equal(got[c], want), "call of %d blocks at %d, ch %d: first differing block %d of %d" % (T, pos, c, int(np.nonzero((got[c] != want).any(axis=1))[0][0]), T)
        st = batch.read_status()
        for c in range(0, n_ch, 3):
            assert int(st["sam_locked"][c]) == orcs[c].getSAMphaseLockStatus(), "lock flag after block %d, ch %d" % (pos + T - 1, c)
            assert np.float32(st["sam_frequency"][c]) == np.float32(orcs[c].getSAMfrequency()), "PLL frequency after block %d, ch %d" % (pos + T - 1, c)
        pos += T
    assert batch.sam_chunk_calls() == sum(1 for T in plan if T >= 16)
    hip.free_all(); batch.close()


def test_in_place_calls_never_take_the_pipeline(gpu, ao):
    """The pipeline's recovery restores channel state, not caller buffers: a call whose output rows alias its I rows (the reference's
    own convention, AudioSDR.cpp:158-165: the audio is written into blockI) keeps the in-kernel block loop -- also with injected
    timeouts pending -- and is exact; the same batch takes the pipeline again for a call with separate buffers."""
    from audiosdr_amd.synth import make_iq
    n_ch, T = 64, 20
    I, Q = make_iq(n_ch, 3 * T, fc=6290.0, A=0.25, noise=0.02, impulse_every=611)
    batch, orcs = _mk(gpu, ao, n_ch, [S("setDemodMode", 1), S("enableAudioFilter"), S("setNoiseBlankerThresholdDb", 10.0)])
    batch.debug_set_stream_spin_limit(1)                        # any pipeline launch would time out and be recovered
    hip = Hip()
    want = [orcs[c].update(I[c], Q[c]).reshape(3 * T, 128) for c in range(n_ch)]
    for k, alias in enumerate(("I", "Q")):
        dI, dQ = hip.upload(I[:, k * T:(k + 1) * T]), hip.upload(Q[:, k * T:(k + 1) * T])
        dO = dI if alias == "I" else dQ
        batch.update_device(dI, dQ, dO, T)
        batch.synchronize()
        got = hip.download(dO, (n_ch, T, 128), np.int16)
        for c in range(n_ch):
            assert np.array_equal(got[c], want[c][k * T:(k + 1) * T]), (alias, c)
    assert batch.stream_pipeline_launches() == 0 and batch.stream_pipeline_recoveries() == 0
    batch.debug_set_stream_spin_limit(0)
    dI, dQ, dO = hip.upload(I[:, 2 * T:]), hip.upload(Q[:, 2 * T:]), hip.malloc(n_ch * T * 256)
    batch.update_device(dI, dQ, dO, T)
    batch.synchronize()
    got = hip.download(dO, (n_ch, T, 128), np.int16)
    for c in range(n_ch):
        assert np.array_equal(got[c], want[c][2 * T:]), c
    assert batch.stream_pipeline_launches() == 1 and batch.stream_pipeline_alloc_failures() == 0
    _compare_status(batch, orcs)
    hip.free_all(); batch.close()


def test_launch_form_switches_per_batch(gpu, ao):
    """asdr_set_stream_pipeline / asdr_set_sam_launch_form: the launch forms are properties of a batch (their defaults come from
    the environment when the batch is created), so two batches of one process can differ -- and every form is bit-exact."""
    from audiosdr_amd.synth import make_iq
    n_ch, T = 64, 12
    fc = 6890.0 + (np.arange(n_ch) % 5 - 2) * 40.0
    I, Q = make_iq(n_ch, T, fc=fc, A=0.3, m=0.4, noise=0.02)
    for mode, kw in ((1, dict(pipeline=True)), (1, dict(pipeline=False)), (5, dict(fused=True)), (5, dict(fused=False, split_min=1))):
        b = gpu.AudioSDRBatch(n_ch)
        b.setDemodMode(mode); b.enableAudioFilter()
        if "pipeline" in kw:
            b.set_stream_pipeline(kw["pipeline"])
        else:
            b.set_sam_launch_form(kw["fused"], kw.get("split_min", 0))
        got = b.update(I, Q)
        assert b.stream_pipeline_launches() == (1 if kw.get("pipeline") else 0)
        if mode == 5:
            assert b.schedule_layout()["sam_three_launches"] == (not kw["fused"])
        for c in range(0, n_ch, 5):
            o = ao.OracleSDR(); o.setDemodMode(mode); o.enableAudioFilter()
            assert np.array_equal(got[c], o.update(I[c], Q[c]).reshape(T, 128)), (mode, kw, c)
        b.close()
