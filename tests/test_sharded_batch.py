"""Sharded batches (include/asdr.h asdr_create_sharded; SURVEY.md 8(e): GPU g of G owns channels [g*C/G, (g+1)*C/G), no collective).

CPU part: control-plane-only shards (ASDR_NO_DEVICE) -- the routing of GLOBAL channel indices through every setter and getter of the
reference's class surface (AudioSDR.h:88-156) must make a sharded batch indistinguishable from one batch, for uneven C / G, for
ASDR_ALL, for getters after per-channel setters.  GPU part (-m gpu): 2 and 3 shards ALL ON DEVICE 0 against the single batch, bit for
bit, on BASELINE config 4's mix (mode = channel mod 7 + ALS notch + blanker at 10 dB), through the host rows (asdr_update: scatter /
gather, one thread per shard), through device pointers, and through the shard handles; status, taps and the capture sink in global
channel order.  Eight GPUs are not available to this suite: the multi-device leg is the same code with other ordinals."""
import numpy as np
import pytest

SETTERS = [
    ("setDemodMode", lambda r: (int(r.integers(0, 7)),)),
    ("setInputGain", lambda r: (float(r.uniform(0.1, 3.0)),)),
    ("setIQgainBalance", lambda r: (float(r.uniform(0.8, 1.2)),)),
    ("setOutputGain", lambda r: (float(r.uniform(0.1, 1.0)),)),
    ("setMute", lambda r: (int(r.integers(0, 2)),)),
    ("enableAudioFilter", lambda r: ()), ("disableAudioFilter", lambda r: ()),
    ("setAudioFilter", lambda r: (int(r.integers(0, 11)),)),
    ("enableALSfilter", lambda r: ()), ("disableALSfilter", lambda r: ()),
    ("setALSfilterNotch", lambda r: ()), ("setALSfilterPeak", lambda r: ()),
    ("setALSfilterAdaptive", lambda r: ()), ("setALSfilterStatic", lambda r: ()),
    ("setALSfilterParams", lambda r: (int(r.integers(1, 130)), float(r.uniform(0.01, 1.0)), float(r.integers(0, 9)))),
    ("enableAGC", lambda r: ()), ("disableAGC", lambda r: ()),
    ("setAGCthreshold", lambda r: (float(r.choice([-60.0, -40.0, -30.0])),)),
    ("setAGCslope", lambda r: (float(r.choice([0.1, 0.25])),)),
    ("setAGCkneeWidth", lambda r: (float(r.choice([2.0, 6.0])),)),
    ("setAGCmode", lambda r: (int(r.integers(0, 4)),)),
    ("setAGCattackTime", lambda r: (float(r.uniform(1, 20)),)), ("setAGCreleaseTime", lambda r: (float(r.uniform(50, 900)),)),
    ("setAGChangTime", lambda r: (float(r.uniform(0, 900)),)), ("setAGCstaticGain", lambda r: (float(r.uniform(1, 20)),)),
    ("enableNoiseBlanker", lambda r: ()), ("disableNoiseBlanker", lambda r: ()),
    ("setNoiseBlankerThreshold", lambda r: (float(r.uniform(1.1, 4.0)),)),
    ("setNoiseBlankerThresholdDb", lambda r: (float(r.uniform(3.0, 20.0)),)),
    ("init", lambda r: ()),
]
GETTERS = ["getMute", "getDemodMode", "getTuningOffset", "getBPFlower", "getBPFupper", "getAudioFilter", "ALSfilterIsEnabled",
           "ALSfilterIsNotch", "ALSfilterIsPeak", "ALSfilterIsAdaptive", "AGCisEnabled", "getAGCthreshold", "getAGCslope",
           "getAGCkneeWidth", "getAGCattack", "getAGCrelease", "getAAGalphaAttack", "getAGCbetaAttack", "getAGCalphaRelease",
           "getAGCbetaRelease", "getAGCstaticGain", "NoiseBlankerisEnabled"]


def _same_control_plane(A, one, sh, n):
    for c in range(n):
        for g in GETTERS:
            a, b = getattr(one, g)(ch=c), getattr(sh, g)(ch=c)
            assert np.float32(a).tobytes() == np.float32(b).tobytes(), (g, c, a, b)
        for i in (0, 1, 64, 100, 128, 129):
            assert np.float32(one.getAGClookup(i, ch=c)).tobytes() == np.float32(sh.getAGClookup(i, ch=c)).tobytes(), (i, c)
        h1, k1 = one.chain_constants(c)
        h2, k2 = sh.chain_constants(c)
        assert h1 == h2 and k1.tobytes() == k2.tobytes()


@pytest.mark.parametrize("n,shards", [(100, 3), (17, 17), (64, 1), (1000, 8)])
def test_global_indices_route_to_the_owner(A, n, shards):
    rng = np.random.default_rng(n * 31 + shards)
    one = A.AudioSDRBatch(n, device=-1)
    sh = A.AudioSDRBatch(n, devices=[-1] * shards)
    assert sh.n_shards == shards and one.n_shards == 1
    from audiosdr_amd.sharding import shard_range
    assert [sh.shard_range(g) for g in range(shards)] == [shard_range(n, g, shards) for g in range(shards)]
    _same_control_plane(A, one, sh, n)
    for step in range(300):
        name, mk = SETTERS[int(rng.integers(0, len(SETTERS)))]
        args = mk(rng)
        ch = A.ALL if rng.random() < 0.15 else int(rng.integers(0, n))
        r1 = getattr(one, name)(*args, ch=ch)
        r2 = getattr(sh, name)(*args, ch=ch)
        if name == "setDemodMode":
            assert r1 == r2          # the tuning offset it returns (of channel 0 for ALL)
    _same_control_plane(A, one, sh, n)
    # out-of-range channels are ignored by both, as ever
    sh.setOutputGain(0.9, ch=n); sh.setOutputGain(0.9, ch=-7)
    assert sh.getAGCthreshold(ch=n) == 0.0 and sh.getDemodMode(ch=n) == 0
    _same_control_plane(A, one, sh, n)
    # what the next update would launch: the shards' schedules add up to the channels
    f1, f2 = one.control_plane_flush(), sh.control_plane_flush()
    assert f2["rows_refilled"] == f1["rows_refilled"] + (shards - 1)       # (each shard carries its own dummy row)
    lay = sh.schedule_layout()
    assert sum(lay[k] for k in ("plain", "sam", "als_long", "als_compact", "sam_als", "remainders")) >= n
    one.close(); sh.close()


def test_shard_handles_and_refusals(A):
    sh = A.AudioSDRBatch(50, devices=[-1, -1])
    v = sh.shard(1)
    assert v.n_channels == 25 and sh.shard_range(1) == (25, 50)
    v.setAGCstaticGain(2.5, ch=3)                    # local index 3 of shard 1 == global channel 28
    assert sh.getAGCstaticGain(ch=28) == 2.5 and sh.getAGCstaticGain(ch=3) == 10.0 and sh.getAGCstaticGain(ch=27) == 10.0
    z = np.zeros((50, 1, 128), np.int16)
    with pytest.raises(A.AsdrError, match="shard 0.*needs a HIP device"):
        sh.update(z, z)
    with pytest.raises(A.AsdrError, match="several devices|shard"):
        sh.update_device(16, 16, 16, 1)
    with pytest.raises(A.AsdrError):
        A.AudioSDRBatch(4, devices=[-1] * 5)         # more shards than channels
    v.close()                                        # a view: must not destroy the shard
    assert sh.getDemodMode(ch=30) == 0
    sh.close()


# ------------------------------------------------------------------------------------------------------------------------------
def _c4_configure(batch, n):
    for m in range(7):
        for c in range(m, n, 7):
            batch.setDemodMode(m, ch=c)
    batch.enableALSfilter()
    batch.setNoiseBlankerThresholdDb(10.0)


def _c4_input(n, nb, seed=0):
    from audiosdr_amd.synth import make_iq
    fc = 6890.0 - 600.0 + 10.0 * (np.arange(n) % 5)
    return make_iq(n, nb, fc=fc, A=0.25, m=0.3, noise=0.02, impulse_every=1900, f2=fc + 1000.0, a2=0.125, seed0=12345 + seed)


@pytest.mark.gpu
@pytest.mark.parametrize("shards", [2, 3])
def test_shards_on_one_device_equal_the_single_batch(gpu, shards):
    n, nb = 1000 + 7 * shards + 3, 5
    I, Q = _c4_input(n, 2 * nb)
    one = gpu.AudioSDRBatch(n)
    sh = gpu.AudioSDRBatch(n, devices=[0] * shards)
    _c4_configure(one, n); _c4_configure(sh, n)
    one.enable_taps(); sh.enable_taps()
    a1 = one.update(I[:, :nb], Q[:, :nb])
    a2 = sh.update(I[:, :nb], Q[:, :nb])                      # host rows: scatter / gather, one thread per shard
    assert np.array_equal(a1, a2)
    # per-channel setters between the calls, by global index
    for c in (0, n // shards - 1, n // shards, n - 1, 511):
        for b in (one, sh):
            b.setOutputGain(0.8, ch=c); b.setDemodMode((c + 3) % 7, ch=c); b.setAGChangTime(0.0, ch=c)
    a1 = one.update(I[:, nb:], Q[:, nb:])
    a2 = sh.update(I[:, nb:], Q[:, nb:])
    assert np.array_equal(a1, a2)
    s1, s2 = one.read_status(), sh.read_status()
    for k in s1:
        assert s1[k].tobytes() == s2[k].tobytes(), k
    t1, t2 = one.read_taps(), sh.read_taps()
    for k in t1:
        assert t1[k].tobytes() == t2[k].tobytes(), k
    for c in (0, 333, 334, n - 1):
        assert one.getSAMphaseLockStatus(ch=c) == sh.getSAMphaseLockStatus(ch=c)
        assert np.float32(one.getAMcarrierLevel(ch=c)).tobytes() == np.float32(sh.getAMcarrierLevel(ch=c)).tobytes()
    one.close(); sh.close()


@pytest.mark.gpu
def test_sharded_device_pointers_capture_and_shard_handles(gpu):
    from tests.helpers import Hip
    n, nb, shards = 803, 9, 3
    I, Q = _c4_input(n, nb, seed=3)
    one = gpu.AudioSDRBatch(n)
    sh = gpu.AudioSDRBatch(n, devices=[0] * shards)
    for b in (one, sh):
        b.setDemodMode(1); b.enableAudioFilter(); b.setDemodMode(4, ch=700); b.setDemodMode(5, ch=17)
    hip = Hip()
    dI, dQ = hip.upload(I), hip.upload(Q)
    d1, d2, d3 = hip.malloc(n * nb * 256), hip.malloc(n * nb * 256), hip.malloc(n * nb * 256)
    one.update_device_strided(dI, dQ, d1, 4, nb, nb); one.update_device_strided(dI + 4 * 256, dQ + 4 * 256, d1 + 4 * 256, nb - 4, nb, nb)
    # (a) global device rows on the sharded handle (all shards on the device that owns the pointers)
    sh.update_device_strided(dI, dQ, d2, 4, nb, nb)
    # (b) then shard by shard, local rows, as a multi-GPU host would (asdr_shard): the remaining blocks
    for g in range(shards):
        lo, hi = sh.shard_range(g)
        v = sh.shard(g)
        off = lo * nb * 256 + 4 * 256
        v.update_device_strided(dI + off, dQ + off, d2 + off, nb - 4, nb, nb)
    sh.synchronize(); one.synchronize()
    w1 = hip.download(d1, (n, nb, 128), np.int16)
    w2 = hip.download(d2, (n, nb, 128), np.int16)
    assert np.array_equal(w1[:, :4], w2[:, :4])
    assert np.array_equal(w1[:, 4:], w2[:, 4:])
    # capture sink through the sharded handle, read back by global channel
    for b in (one, sh):
        b.capture_open(16)
        b.capture_update_device(dI, dQ, 3, in_stride_blocks=nb)
    assert sh.capture_position == 3 and sh.capture_capacity == 16
    for c in (0, 267, 268, 700, n - 1):
        assert np.array_equal(one.capture_read(c), sh.capture_read(c)), c
    hip.free_all(); one.close(); sh.close()


# ------------------------------------------------------------------------------------------------------------------------------
# The overlapped host path's chunk plan (asdr_update): host logic, checked without a device.
@pytest.mark.parametrize("kind", ["one_group", "c4_mix", "halves", "scattered"])
@pytest.mark.parametrize("chunks", [1, 2, 5, 16])
def test_host_plan_never_runs_ahead_of_its_data(A, kind, chunks):
    """For every settings mix and chunk count: (1) the kernel parts of a call launch every schedule slot exactly once; (2) every channel
    a part touches lies in an input chunk <= need_in[part] (so its rows have arrived when the part starts: parts are enqueued in order
    behind the event of chunk need_in[part]); (3) every channel of output chunk j is touched by a part <= last_part[j] (so the chunk
    is complete when it is copied out); (4) for a batch of one settings group the plan is the perfect pipeline need_in[p] = p =
    last_part[p]."""
    import ctypes as C
    rng = np.random.default_rng(7)
    n = 5003
    b = A.AudioSDRBatch(n, device=-1)
    L = A.load_library()
    if kind == "one_group":
        b.setDemodMode(1); b.enableAudioFilter()
    elif kind == "c4_mix":
        for c in range(n):
            L.asdr_setDemodMode(b._h, c, c % 7)
        b.enableALSfilter()
    elif kind == "halves":
        b.setDemodMode(1)
        for c in range(n // 2, n):
            L.asdr_setDemodMode(b._h, c, 5)
        for c in range(100, 150):
            b.enableALSfilter(ch=c); b.setALSfilterParams(100, 0.3, 5, ch=c)
    else:
        for c in range(n):
            L.asdr_setDemodMode(b._h, c, int(rng.integers(0, 7)))
            if rng.random() < 0.3:
                L.asdr_enableALSfilter(b._h, c)
            if rng.random() < 0.2:
                L.asdr_enableAudioFilter(b._h, c)
    b.control_plane_flush()
    for name, at, rt in (("asdr_debug_host_plan", [C.c_void_p, C.c_int] + [C.POINTER(C.c_int)] * 3, C.c_int),
                         ("asdr_debug_part_slots", [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int)], C.c_int),
                         ("asdr_debug_schedule", [C.c_void_p, C.POINTER(C.c_int), C.c_int], C.c_int)):
        getattr(L, name).argtypes = at; getattr(L, name).restype = rt
    K = chunks
    bound, need_in, last_part = (C.c_int * (K + 1))(), (C.c_int * K)(), (C.c_int * K)()
    assert L.asdr_debug_host_plan(b._h, K, bound, need_in, last_part) == 0, L.asdr_last_error()
    bound, need_in, last_part = list(bound), list(need_in), list(last_part)
    assert bound[0] == 0 and bound[-1] == n and all(bound[j] < bound[j + 1] for j in range(K))
    sched = (C.c_int * (n + 64))()
    n_slots = L.asdr_debug_schedule(b._h, sched, n + 64)
    assert n <= n_slots <= n + 64
    sched = np.array(sched[:n_slots])
    chunk_of = np.searchsorted(np.array(bound[1:]), np.arange(n), side="right")          # channel -> chunk
    seen = np.zeros(n_slots, dtype=int)
    for p in range(K):
        out = (C.c_int * 32)()
        m = L.asdr_debug_part_slots(b._h, p, K, out)
        assert m >= 0
        for i in range(m):
            first, cnt = out[2 * i], out[2 * i + 1]
            assert first % 8 == 0 and cnt % 8 == 0
            seen[first:first + cnt] += 1
            chans = sched[first:first + cnt]
            chans = chans[chans < n]
            if len(chans):
                assert chunk_of[chans].max() <= need_in[p], (p, int(chunk_of[chans].max()), need_in[p])      # (2)
                for j in np.unique(chunk_of[chans]):
                    assert last_part[j] >= p, (j, last_part[j], p)                                            # (3)
    assert (seen == 1).all()                                                                                  # (1)
    assert sorted(sched[sched < n].tolist()) == list(range(n))
    assert all(need_in[p] <= need_in[p + 1] for p in range(K - 1))
    if kind == "one_group":
        # (8-channel waves: a part's last wave may reach into the next chunk by up to 7 channels, never further)
        assert all(need_in[p] in (p, min(p + 1, K - 1)) for p in range(K)) and all(last_part[j] in (j, max(j - 1, 0), min(j + 1, K - 1)) for j in range(K))
    b.close()
