"""Randomised and long-running parity: seeded fuzz over the whole control surface, decay into float32 denormals,
pathological inputs.  Everything bit-exact against the oracle, through the C ABI."""
import numpy as np
import pytest

from helpers import Hip, S, apply_setters, compare_status

pytestmark = pytest.mark.gpu


def _random_setter(rng):
    k = rng.integers(0, 24)
    f = float
    table = [
        lambda: S("setDemodMode", int(rng.integers(0, 7))),
        lambda: S("setDemodMode", int(rng.integers(0, 7))),
        lambda: S("enableAudioFilter"), lambda: S("disableAudioFilter"),
        lambda: S("setAudioFilter", int(rng.integers(0, 11))),
        lambda: S("enableNoiseBlanker"), lambda: S("disableNoiseBlanker"),
        lambda: S("setNoiseBlankerThresholdDb", f(rng.choice([3.0, 6.0, 10.0, 20.0]))),
        lambda: S("setNoiseBlankerThreshold", f(rng.choice([1.2, 2.0, 5.0]))),
        lambda: S("enableAGC"), lambda: S("disableAGC"), lambda: S("setAGCmode", int(rng.integers(0, 4))),
        lambda: S("setAGCthreshold", f(rng.choice([-60.0, -40.0, -20.0]))), lambda: S("setAGCslope", f(rng.choice([0.1, 0.3, 0.7]))),
        lambda: S("setAGCkneeWidth", f(rng.choice([2.0, 6.0]))), lambda: S("setAGCstaticGain", f(rng.choice([1.0, 10.0, 30.0]))),
        lambda: S("setAGChangTime", f(rng.choice([1.0, 20.0, 100.0]))),
        lambda: S("enableALSfilter"), lambda: S("disableALSfilter"),
        lambda: S(str(rng.choice(["setALSfilterNotch", "setALSfilterPeak", "setALSfilterAdaptive", "setALSfilterStatic"]))),
        lambda: S("setALSfilterParams", int(rng.choice([8, 32, 55, 100, 128])), f(rng.choice([0.05, 0.5])), f(rng.choice([1.0, 3.0, 9.0]))),
        lambda: S("setInputGain", f(rng.choice([0.5, 1.0, 2.5]))), lambda: S("setIQgainBalance", f(rng.choice([0.95, 1.02]))),
        lambda: S("setOutputGain", f(rng.choice([0.25, 0.5, 1.0]))),
    ]
    return table[k]()


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12])
def test_fuzz_control_surface(gpu, ao, seed):
    """40 channels x 36 blocks: every channel starts from a random configuration and receives random setter calls
    between blocks (each call applied to a random subset of channels, identically on both sides)."""
    from audiosdr_amd.synth import make_iq
    rng = np.random.default_rng(seed)
    n_ch, n_blk = 40, 36
    fc = 6890.0 + rng.uniform(-1800, 1800, n_ch)
    I, Q = make_iq(n_ch, n_blk, fc=fc, A=rng.uniform(0.01, 0.6, n_ch), m=0.4, fm=300.0, impulse_every=int(rng.integers(300, 900)),
                   f2=fc + 700.0, a2=0.05)
    batch = gpu.AudioSDRBatch(n_ch)
    orcs = [ao.OracleSDR() for _ in range(n_ch)]
    for _ in range(60):                                   # random initial configuration
        meth, args, _sel = _random_setter(rng)
        mask = rng.random(n_ch) < 0.3
        apply_setters(batch, orcs, [S(meth, *args, sel=lambda c, m=mask: bool(m[c]))])
    for b in range(n_blk):
        for _ in range(int(rng.integers(0, 4))):
            meth, args, _sel = _random_setter(rng)
            mask = rng.random(n_ch) < 0.2
            apply_setters(batch, orcs, [S(meth, *args, sel=lambda c, m=mask: bool(m[c]))])
        got = batch.update(I[:, b:b + 1], Q[:, b:b + 1])[:, 0]
        for c in range(n_ch):
            want = orcs[c].update(I[c, b], Q[c, b])
            assert np.array_equal(got[c], want), "seed %d block %d ch %d (mode %d)" % (seed, b, c, orcs[c].getDemodMode())
    compare_status(gpu, batch, orcs)
    batch.close()


@pytest.mark.parametrize("seed", [21, 22, 23, 24])
def test_fuzz_whole_waves(gpu, ao, seed):
    """The same fuzz with the channels configured in groups of 8 (48 groups x 8 channels x 24 blocks): whole waves of one schedule
    key, i.e. the uniform-key instantiations of every kernel kind (plain, SAM, ALS on the compact and on the long rows), with
    the groups moving between kinds mid-stream.  (With 40 individually configured channels nearly every channel is a remainder
    and runs in the general kernel.)"""
    from audiosdr_amd.synth import make_iq
    rng = np.random.default_rng(seed)
    n_grp, n_blk = 48, 24
    n_ch = 8 * n_grp
    fc = 6890.0 + rng.uniform(-1800, 1800, n_ch)
    I, Q = make_iq(n_ch, n_blk, fc=fc, A=rng.uniform(0.01, 0.6, n_ch), m=0.4, fm=300.0, impulse_every=int(rng.integers(300, 900)),
                   f2=fc + 700.0, a2=0.05)
    batch = gpu.AudioSDRBatch(n_ch)
    orcs = [ao.OracleSDR() for _ in range(n_ch)]
    for _ in range(80):
        meth, args, _sel = _random_setter(rng)
        mask = rng.random(n_grp) < 0.3
        apply_setters(batch, orcs, [S(meth, *args, sel=lambda c, m=mask: bool(m[c // 8]))])
    for b in range(n_blk):
        for _ in range(int(rng.integers(0, 4))):
            meth, args, _sel = _random_setter(rng)
            mask = rng.random(n_grp) < 0.2
            apply_setters(batch, orcs, [S(meth, *args, sel=lambda c, m=mask: bool(m[c // 8]))])
        got = batch.update(I[:, b:b + 1], Q[:, b:b + 1])[:, 0]
        for c in range(n_ch):
            want = orcs[c].update(I[c, b], Q[c, b])
            assert np.array_equal(got[c], want), "seed %d block %d ch %d (mode %d)" % (seed, b, c, orcs[c].getDemodMode())
    compare_status(gpu, batch, orcs)
    batch.close()


@pytest.mark.parametrize("seed", [31, 32])
def test_fuzz_large_mixed_batch(gpu, ao, seed):
    """4,000 channels with random modes, enables and ALS filter shapes (both sides of the compact layout's limits), enough SAM
    channels for the three-launch SAM path, calls of 1-3 blocks, parameters changing between calls: every kernel kind in whole waves
    AND the remainders' sub-range in one schedule, three times rebuilt.  Every channel against the oracle."""
    from audiosdr_amd.synth import make_iq
    rng = np.random.default_rng(seed)
    n_ch = 4000
    calls = [int(x) for x in rng.integers(1, 4, 4)]
    n_blk = sum(calls)
    fc = 6890.0 + rng.uniform(-1500, 1500, n_ch)
    I, Q = make_iq(n_ch, n_blk, fc=fc, A=rng.uniform(0.05, 0.5, n_ch), m=0.4, fm=300.0, impulse_every=int(rng.integers(400, 900)), f2=fc + 600.0, a2=0.08)
    batch = gpu.AudioSDRBatch(n_ch)
    orcs = [ao.OracleSDR() for _ in range(n_ch)]
    shapes = [(55, 0.5, 3.0), (64, 0.5, 1.0), (65, 0.25, 0.0), (32, 0.5, 33.0), (32, 0.5, 34.0), (100, 0.05, 7.0), (8, 0.5, 0.0), (128, 0.05, 1.0)]
    def reconfigure(frac):
        grp = rng.integers(0, 24, n_ch)            # 24 configuration groups of ~167 channels: whole waves + a remainder each
        pick = rng.random(24) < frac
        for g in np.nonzero(pick)[0]:
            sel = lambda c, g=g: grp[c] == g
            st = [S("setDemodMode", int(rng.choice([0, 1, 2, 4, 5, 5, 5, 6]))), S("setNoiseBlankerThresholdDb", 10.0)]
            if rng.random() < 0.7:
                st += [S("enableALSfilter"), S("setALSfilterParams", *shapes[int(rng.integers(0, len(shapes)))])]
                if rng.random() < 0.3: st.append(S("setALSfilterPeak"))
                if rng.random() < 0.2: st.append(S("setALSfilterStatic"))
            else:
                st.append(S("disableALSfilter"))
            if rng.random() < 0.5: st.append(S("enableAudioFilter"))
            apply_setters(batch, orcs, [S(m, *a, sel=sel) for (m, a, _s) in st])
    reconfigure(1.0)
    b0 = 0
    for T in calls:
        got = batch.update(I[:, b0:b0 + T], Q[:, b0:b0 + T])
        for c in range(n_ch):
            want = orcs[c].update(I[c, b0:b0 + T], Q[c, b0:b0 + T]).reshape(T, 128)
            assert np.array_equal(got[c], want), "seed %d blocks %d.. ch %d (mode %d)" % (seed, b0, c, orcs[c].getDemodMode())
        b0 += T
        reconfigure(0.3)
    compare_status(gpu, batch, orcs)
    batch.close()


@pytest.mark.parametrize("mode", [1, 4, 5])
def test_decay_into_denormals(gpu, ao, mode):
    """Signal for 12 blocks, then digital silence for 400 blocks: biquad, AGC, blanker-average and PLL states decay
    through the float32 denormal range (the CPU reference keeps denormals; so must the GPU), then signal again."""
    from audiosdr_amd.synth import make_iq
    I, Q = make_iq(3, 12, fc=6890.0 if mode != 1 else 6290.0, A=0.3, m=0.5)
    Z = np.zeros((3, 400, 128), np.int16)
    I = np.concatenate([I, Z, I], axis=1); Q = np.concatenate([Q, Z, Q], axis=1)
    batch = gpu.AudioSDRBatch(3)
    orcs = [ao.OracleSDR() for _ in range(3)]
    apply_setters(batch, orcs, [S("setDemodMode", mode), S("enableAudioFilter"), S("setNoiseBlankerThresholdDb", 10.0),
                                S("disableAGC", sel=lambda c: c == 1), S("enableALSfilter", sel=lambda c: c == 2)])
    got = batch.update(I, Q)
    want = np.stack([orcs[c].update(I[c], Q[c]).reshape(-1, 128) for c in range(3)])
    assert np.array_equal(got, want)
    compare_status(gpu, batch, orcs)
    batch.close()


def test_pathological_inputs(gpu, ao):
    """Full-scale square waves, Nyquist alternation, DC, single-sample spikes, -32768."""
    n_blk = 10
    t = np.arange(n_blk * 128)
    pats = [
        (np.where((t // 3) % 2 == 0, 32767, -32768), np.where((t // 5) % 2 == 0, -32768, 32767)),
        (np.where(t % 2 == 0, 32767, -32767), np.where(t % 2 == 0, -32767, 32767)),
        (np.full_like(t, 12345), np.full_like(t, -23456)),
        (np.where(t % 257 == 0, 32767, 0), np.where(t % 263 == 0, -32768, 0)),
        (np.full_like(t, -32768), np.full_like(t, -32768)),
    ]
    I = np.stack([p[0] for p in pats]).astype(np.int16).reshape(len(pats), n_blk, 128)
    Q = np.stack([p[1] for p in pats]).astype(np.int16).reshape(len(pats), n_blk, 128)
    for mode in (0, 3, 4, 5, 6):
        batch = gpu.AudioSDRBatch(len(pats))
        orcs = [ao.OracleSDR() for _ in pats]
        apply_setters(batch, orcs, [S("setDemodMode", mode), S("enableAudioFilter"), S("enableALSfilter"), S("setInputGain", 4.0),
                                    S("setOutputGain", 1.0)])
        got = batch.update(I, Q)
        want = np.stack([orcs[c].update(I[c], Q[c]).reshape(-1, 128) for c in range(len(pats))])
        assert np.array_equal(got, want), mode
        compare_status(gpu, batch, orcs)
        batch.close()


@pytest.mark.parametrize("seed,ssb_only", [(101, True), (102, True), (103, False), (104, False)])
def test_fuzz_multi_block_calls(gpu, ao, seed, ssb_only):
    """64 channels in groups of 8 that always share their configuration (so the waves keep one schedule key), random setter calls
    between calls of 1 / 3 / 8 / 17 / 30 blocks.  With SSB-class and AM modes only, the calls of 8 blocks and more run as the streaming
    block pipeline (groups change between SSB and AM role sets mid-stream); with all modes they run block by block or through the in-kernel block loop.  Every block against the oracle."""
    from audiosdr_amd.synth import make_iq
    rng = np.random.default_rng(seed)
    n_ch, plan = 64, [int(rng.choice([1, 3, 8, 17, 30])) for _ in range(7)]
    total = sum(plan)
    grp_fc = 6890.0 + rng.uniform(-1500, 1500, n_ch // 8)
    fc = np.repeat(grp_fc, 8)
    I, Q = make_iq(n_ch, total, fc=fc, A=np.repeat(rng.uniform(0.02, 0.5, n_ch // 8), 8), m=0.3, fm=250.0,
                   impulse_every=int(rng.integers(300, 900)))
    batch = gpu.AudioSDRBatch(n_ch)
    orcs = [ao.OracleSDR() for _ in range(n_ch)]

    def random_group_setters(k):
        out = []
        for _ in range(k):
            meth, args, _sel = _random_setter(rng)
            if ssb_only and meth == "setDemodMode":
                args = (int(rng.choice([0, 1, 2, 3, 4, 6])),)   # the modes the block pipeline has role sets for (SSB class + AM)
            if meth in ("enableALSfilter", "setALSfilterParams") and ssb_only:
                continue
            groups = rng.random(n_ch // 8) < 0.35
            out.append(S(meth, *args, sel=lambda c, g=groups: bool(g[c // 8])))
        return out

    apply_setters(batch, orcs, [S("setDemodMode", 1), S("disableALSfilter")] + random_group_setters(40))
    hip = Hip()
    pos = 0
    for T in plan:
        apply_setters(batch, orcs, random_group_setters(int(rng.integers(0, 3))))
        dI, dQ = hip.upload(I[:, pos:pos + T]), hip.upload(Q[:, pos:pos + T])
        dO = hip.malloc(n_ch * T * 256)
        batch.update_device(dI, dQ, dO, T)
        batch.synchronize()
        got = hip.download(dO, (n_ch, T, 128), np.int16)
        for c in range(n_ch):
            want = orcs[c].update(I[c, pos:pos + T], Q[c, pos:pos + T]).reshape(T, 128)
            assert np.array_equal(got[c], want), "seed %d call of %d blocks at %d, ch %d (mode %d)" % (seed, T, pos, c, orcs[c].getDemodMode())
        pos += T
    if ssb_only:
        assert batch.stream_pipeline_launches() == sum(1 for T in plan if T >= 8)
    compare_status(gpu, batch, orcs)
    hip.free_all(); batch.close()
