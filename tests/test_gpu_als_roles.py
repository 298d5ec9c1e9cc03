"""ALS role streams (include/asdr.h, asdr_host.cpp): a SMALL bank whose schedule is one settings group of channels with a short ALS filter runs
a multi-block call as chain | filter launches per block on two event-chained streams, the filter of block b beside the chain of block b + 1,
through a stage of post-AGC rows (and, in the three-stage form, tile sets between the chain's halves), a chunk of 8 blocks per launch.  Everything against the oracle, bit for bit; the form must leave the als_x ring as the fused
kernel expects it (single-block calls in between), honour setters between calls, and keep off in-place calls."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _bank(gpu, ao, n_ch, cfg):
    b = gpu.AudioSDRBatch(n_ch)
    cfg(b)
    orcs = []
    for _ in range(n_ch):
        o = ao.OracleSDR(); cfg(o); orcs.append(o)
    return b, orcs


@pytest.mark.parametrize("stages", [3, 2])
@pytest.mark.parametrize("mode,params", [(1, None), (0, (64, 0.5, 1)), (4, (33, 0.7, 32))])
def test_als_role_streams_equal_the_oracle(gpu, ao, mode, params, stages, monkeypatch):
    monkeypatch.setenv("ASDR_ALS_ROLE_STAGES", str(stages))   # (read at asdr_create: 3 = front half | back half | filter on three streams, 2 = chain | filter)
    import torch
    from audiosdr_amd.synth import make_iq
    n_ch, plan = 72, (1, 6, 1, 5, 2, 9)
    total = sum(plan)
    I, Q = make_iq(n_ch, total, fc=6890.0 + (np.arange(n_ch) % 9 - 4) * 35.0, A=0.3, m=0.4, f2=7600.0, a2=0.12, noise=0.01, impulse_every=700)

    def cfg(s):
        s.setDemodMode(mode); s.setNoiseBlankerThresholdDb(10.0); s.enableAudioFilter(); s.enableALSfilter()
        if params is not None:
            s.setALSfilterParams(*params)

    b, orcs = _bank(gpu, ao, n_ch, cfg)
    dI = torch.from_numpy(I).cuda(); dQ = torch.from_numpy(Q).cuda()
    dO = torch.zeros((n_ch, total, 128), dtype=torch.int16, device="cuda")
    torch.cuda.synchronize()
    pos, role_calls = 0, 0
    want = [[] for _ in range(n_ch)]
    for k, T in enumerate(plan):
        if k == 3:   # a setter between calls: the filter's step size (no change of the kernel kind)
            b.setALSfilterParams(*( (55, 0.25, 3) if params is None else (params[0], 0.25, params[2]) ))
            for o in orcs:
                o.setALSfilterParams(*( (55, 0.25, 3) if params is None else (params[0], 0.25, params[2]) ))
        b.update_device_strided(dI.data_ptr() + pos * 256, dQ.data_ptr() + pos * 256, dO.data_ptr() + pos * 256, T, total, total)
        b.synchronize()
        role_calls += 1 if T >= 2 else 0
        if not os.environ.get("ASDR_NO_ALS_ROLE_STREAMS"):
            assert b.als_role_calls() == role_calls, "call %d (%d blocks): the role streams %s" % (k, T, "were not used" if T >= 2 else "ran for one block")
        for c in range(n_ch):
            want[c].append(orcs[c].update(I[c, pos:pos + T], Q[c, pos:pos + T]).reshape(T, 128))
        pos += T
    got = dO.cpu().numpy()
    for c in range(n_ch):
        w = np.concatenate(want[c])
        assert np.array_equal(got[c], w), "channel %d: first differing block %d" % (c, int(np.nonzero((got[c] != w).any(axis=1))[0][0]))
    b.close()


@pytest.mark.parametrize("stages", [3, 2])
def test_als_role_streams_long_calls_wrap_every_ring(gpu, ao, stages, monkeypatch):
    """Calls of 40, 1, 70 and 9 blocks: 70 blocks = 9 chunks, i.e. more than the stage's 32 slots and the 32 tile sets, so
    the back-pressure wait (chain chunk k behind filter chunk k - 3), the stage ring wrapping onto the seeded slot 31, the front stage's wait for
    the back stage (k >= 4) and the tile-set wrap are all exercised -- and compared bit for bit, every channel, every block."""
    monkeypatch.setenv("ASDR_ALS_ROLE_STAGES", str(stages))
    import torch
    from audiosdr_amd.synth import make_iq
    n_ch, plan = 40, (40, 1, 70, 9)
    total = sum(plan)
    I, Q = make_iq(n_ch, total, fc=6890.0 + (np.arange(n_ch) % 9 - 4) * 35.0, A=0.3, m=0.4, f2=7600.0, a2=0.12, noise=0.01, impulse_every=900)

    def cfg(s):
        s.setDemodMode(1); s.setNoiseBlankerThresholdDb(10.0); s.enableAudioFilter(); s.enableALSfilter()

    b, orcs = _bank(gpu, ao, n_ch, cfg)
    dI = torch.from_numpy(I).cuda(); dQ = torch.from_numpy(Q).cuda()
    dO = torch.zeros((n_ch, total, 128), dtype=torch.int16, device="cuda")
    torch.cuda.synchronize()
    pos = 0
    for T in plan:
        b.update_device_strided(dI.data_ptr() + pos * 256, dQ.data_ptr() + pos * 256, dO.data_ptr() + pos * 256, T, total, total)
        b.synchronize()
        pos += T
    if not os.environ.get("ASDR_NO_ALS_ROLE_STREAMS"):
        assert b.als_role_calls() == sum(1 for T in plan if T >= 2)
    got = dO.cpu().numpy()
    for c in range(n_ch):
        w = orcs[c].update(I[c], Q[c]).reshape(total, 128)
        assert np.array_equal(got[c], w), "channel %d: first differing block %d" % (c, int(np.nonzero((got[c] != w).any(axis=1))[0][0]))
    b.close()


def test_als_role_streams_keep_off_in_place_calls_and_mixed_schedules(gpu, ao):
    import torch
    from audiosdr_amd.synth import make_iq
    n_ch, T = 24, 4
    I, Q = make_iq(n_ch, 2 * T, fc=6290.0, A=0.3, m=0.4, f2=7600.0, a2=0.12, noise=0.01)

    def cfg(s):
        s.setDemodMode(1); s.enableALSfilter()

    b, orcs = _bank(gpu, ao, n_ch, cfg)
    dI = torch.from_numpy(I).cuda(); dQ = torch.from_numpy(Q).cuda()
    torch.cuda.synchronize()
    # in place: the output rows are the I rows (the reference's own convention) -> the block loop
    b.update_device_strided(dI.data_ptr(), dQ.data_ptr(), dI.data_ptr(), T, 2 * T, 2 * T)
    b.synchronize()
    assert b.als_role_calls() == 0
    got = dI.cpu().numpy()[:, :T]
    want1 = [o.update(I[c, :T], Q[c, :T]).reshape(T, 128) for c, o in enumerate(orcs)]
    # a second settings group in the schedule (one channel in another mode) -> not one sub-range -> the block loop
    b.setDemodMode(0, ch=3); orcs[3].setDemodMode(0)
    dO = torch.zeros((n_ch, T, 128), dtype=torch.int16, device="cuda")
    b.update_device_strided(dI.data_ptr() + T * 256, dQ.data_ptr() + T * 256, dO.data_ptr(), T, 2 * T, T)
    b.synchronize()
    assert b.als_role_calls() == 0
    got2 = dO.cpu().numpy()
    for c in range(n_ch):
        want2 = orcs[c].update(I[c, T:], Q[c, T:]).reshape(T, 128)
        assert np.array_equal(got[c], want1[c]) and np.array_equal(got2[c], want2), c
    b.close()


def test_als_role_streams_on_the_shards_of_a_sharded_batch(gpu, ao):
    """Two shards on device 0: device rows of a sharded batch go shard by shard; each shard's schedule is one short-filter settings group, so
    each takes the role streams (they share the device's stream pool: the calls serialise, the results do not change)."""
    import torch
    from audiosdr_amd.synth import make_iq
    n_ch, T = 48, 10

    def cfg(s):
        s.setDemodMode(3); s.setNoiseBlankerThresholdDb(10.0); s.enableALSfilter()

    I, Q = make_iq(n_ch, T, fc=6890.0 + (np.arange(n_ch) % 5 - 2) * 40.0, A=0.3, m=0.4, f2=7600.0, a2=0.12, noise=0.01)
    sh = gpu.AudioSDRBatch(n_ch, devices=[0, 0])
    cfg(sh)
    orcs = []
    for _ in range(n_ch):
        o = ao.OracleSDR(); cfg(o); orcs.append(o)
    dI = torch.from_numpy(I).cuda(); dQ = torch.from_numpy(Q).cuda()
    dO = torch.zeros((n_ch, T, 128), dtype=torch.int16, device="cuda")
    torch.cuda.synchronize()
    sh.update_device_strided(dI.data_ptr(), dQ.data_ptr(), dO.data_ptr(), T, T, T)
    sh.synchronize()
    assert sh.als_role_calls() == 2
    got = dO.cpu().numpy()
    for c in range(n_ch):
        assert np.array_equal(got[c], orcs[c].update(I[c], Q[c]).reshape(T, 128)), c
    sh.close()
