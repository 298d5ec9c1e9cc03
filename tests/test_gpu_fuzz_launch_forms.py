"""Randomised equivalence of the round-4 launch forms.  Whatever the sequence of calls and setters, a batch that uses everything --
shards (1-3, all on device 0), lanes (calls on ASDR_STREAM_BATCH, any lane count), the overlapped host path with any chunk count,
multi-block calls, the block pipeline where it applies -- must produce exactly what a plain batch produces that uses none of it
(one batch, lanes off, device rows on one stream, one block per call): same kernels, same channels, only the launch geometry differs.
The plain form itself is the one every other GPU test holds against the oracle; a few channels are checked against it here as well."""
import numpy as np
import pytest

from tests.helpers import Hip

pytestmark = pytest.mark.gpu

MODES = (0, 1, 2, 3, 4, 5, 6)


def _random_settings(rng, n, big_groups):
    """[(method, args, channels)]: settings groups of very different sizes, so that whole-wave sub-ranges of several kernel kinds,
    remainders and single odd channels all occur."""
    ops = []
    if big_groups:
        cuts = sorted(rng.choice(np.arange(1, n), size=min(4, n - 1), replace=False).tolist())
        ranges = list(zip([0] + cuts, cuts + [n]))
    else:
        ranges = [(0, n)]
    for lo, hi in ranges:
        chans = list(range(lo, hi))
        ops.append(("setDemodMode", (int(rng.choice(MODES)),), chans))
        if rng.random() < 0.5:
            ops.append(("enableAudioFilter", (), chans))
        if rng.random() < 0.3:
            ops.append(("enableALSfilter", (), chans))
        if rng.random() < 0.3:
            ops.append(("setNoiseBlankerThresholdDb", (float(rng.choice([6.0, 10.0])),), chans))
        if rng.random() < 0.2:
            ops.append(("disableNoiseBlanker", (), chans))
    for _ in range(int(rng.integers(0, 6))):      # odd single channels
        c = int(rng.integers(0, n))
        ops.append((str(rng.choice(["setDemodMode", "setAGCmode"])), (int(rng.integers(0, 4)),), [c]))
    return ops


def _apply(batch, ops, L=None):
    for meth, args, chans in ops:
        if len(chans) == batch.n_channels:
            getattr(batch, meth)(*args)
        else:
            for c in chans:
                getattr(batch, meth)(*args, ch=c)


def _seeds():
    import os
    e = os.environ.get("ASDR_FUZZ_FORMS_SEEDS")      # "first-last": more seeds than the suite's (tools/fuzz_more.py style runs)
    if e:
        a, b = e.split("-")
        return list(range(int(a), int(b) + 1))
    return [1, 2, 3, 4, 5, 6, 7, 8]


def run_sequence(gpu, seed, sync_each=False, log=None):
    """One random sequence; returns nothing, asserts equality.  sync_each: synchronise and compare after every call (the replay tool:
    a seed that fails without and passes with it is an ordering bug between launch forms); log: a callable for one line per step."""
    from audiosdr_amd.synth import make_iq
    log = log or (lambda *a: None)
    rng = np.random.default_rng(1000 + seed)
    # sizes around the thresholds: the pipeline (<= 512 groups), the lanes (>= 1024 waves), one launch per block (>= 1024 waves)
    n = int(rng.choice([40, 700, 4100, 8200 + int(rng.integers(0, 9)), 16384 + 5]))
    uniq = min(n, 64)
    total = int(rng.integers(14, 30))
    fc = 6890.0 - 400.0 + 30.0 * (np.arange(uniq) % 7)
    bI, bQ = make_iq(uniq, total, fc=fc, A=0.3, m=0.4, noise=0.02, impulse_every=int(rng.choice([0, 1500, 4000])), f2=fc + 700.0, a2=0.1)
    reps = (n + uniq - 1) // uniq
    I = np.ascontiguousarray(np.tile(bI, (reps, 1, 1))[:n]); Q = np.ascontiguousarray(np.tile(bQ, (reps, 1, 1))[:n])
    shards = int(rng.integers(1, 4))
    subj = gpu.AudioSDRBatch(n, devices=[0] * shards) if shards > 1 else gpu.AudioSDRBatch(n)
    plain = gpu.AudioSDRBatch(n)
    plain.set_lanes(False); plain.set_stream_pipeline(False)
    lanes = int(rng.choice([1, 2, 3, 4]))
    subj.set_lanes(lanes, 64)                       # lanes from 64 waves on: also on the small sizes
    ops = _random_settings(rng, n, big_groups=bool(rng.integers(0, 2)))
    _apply(subj, ops); _apply(plain, ops)
    log("n", n, "blocks", total, "shards", shards, "lanes", lanes, "settings", [(o[0], o[1], len(o[2])) for o in ops])
    hip = Hip()
    s1 = hip.stream()
    dI, dQ = hip.upload(I), hip.upload(Q)
    dI2 = hip.upload(I)                             # the subject's own copy of the I rows: some calls write their audio over them (in place)
    dS, dP = hip.malloc(n * total * 256), hip.malloc(n * total * 256)
    pos, host_rows, in_place, taps_on = 0, {}, [], False
    while pos < total:
        T = int(min(total - pos, rng.choice([1, 1, 1, 2, 3, 9, 12])))
        off = pos * 256
        form = str(rng.choice(["batch", "batch", "stream", "null", "host", "host_pinned"]))
        if form in ("host", "host_pinned"):
            subj.set_host_chunks(int(rng.choice([0, 1, 2, 3, 5])))
            if form == "host_pinned":
                hI, hQ, hO = (gpu.host_alloc((n, T, 128)) for _ in range(3))
                hI[:] = I[:, pos:pos + T]; hQ[:] = Q[:, pos:pos + T]
                subj.update_into(hI, hQ, hO)
                host_rows[pos] = hO.copy()
                for a in (hI, hQ, hO):
                    gpu.host_free(a)
            else:
                host_rows[pos] = subj.update(I[:, pos:pos + T], Q[:, pos:pos + T])
        elif shards > 1 and rng.random() < 0.5:     # shard by shard, local rows, as a multi-GPU host would
            for g in range(shards):
                lo, hi = subj.shard_range(g)
                o2 = off + lo * total * 256
                subj.shard(g).update_device_strided(dI + o2, dQ + o2, dS + o2, T, total, total, gpu.STREAM_BATCH if form == "batch" else (s1 if form == "stream" else 0))
        else:
            st = gpu.STREAM_BATCH if (form == "batch" and shards == 1) else (s1 if form == "stream" else 0)
            if rng.random() < 0.25:                 # in place, the reference's own convention (the audio goes into blockI: AudioSDR.cpp:158-165)
                subj.update_device_strided(dI2 + off, dQ + off, dI2 + off, T, total, total, st)
                in_place.append((pos, T))
            else:
                subj.update_device_strided(dI + off, dQ + off, dS + off, T, total, total, st)
        for k in range(T):                          # the plain form: one block per call, one stream
            plain.update_device_strided(dI + off + k * 256, dQ + off + k * 256, dP + off + k * 256, 1, total, total, 0)
        log("call at block", pos, "T", T, form, "in place" if (in_place and in_place[-1][0] == pos) else "", "taps" if taps_on else "",
            "| pipeline launches", subj.stream_pipeline_launches(), "lane calls", subj.lane_calls())
        if sync_each:
            subj.synchronize(); plain.synchronize()
            if taps_on:
                tS, tP = subj.read_taps(), plain.read_taps()
                log("   taps differ:", [k for k in tS if tS[k].tobytes() != tP[k].tobytes()])
        pos += T
        if rng.random() < 0.35:                     # setters between calls, by global channel index
            c = int(rng.integers(0, n))
            meth, args = [("setOutputGain", (float(rng.uniform(0.2, 1.0)),)), ("setDemodMode", (int(rng.choice(MODES)),)),
                          ("setAGChangTime", (float(rng.choice([0.0, 50.0])),)), ("enableALSfilter", ()), ("disableALSfilter", ()),
                          ("setMute", (int(rng.integers(0, 2)),))][int(rng.integers(0, 6))]
            for b in (subj, plain):
                getattr(b, meth)(*args, ch=c)
            log("   setter", meth, args, "ch", c)
        if rng.random() < 0.15:
            subj.read_status()                      # a host-side read in between (synchronises the lanes)
        if rng.random() < 0.1:                      # stage taps on / off (while on: no lanes, no pipeline)
            taps_on = not taps_on
            subj.enable_taps(taps_on); plain.enable_taps(taps_on)
            log("   taps", taps_on)
    subj.synchronize(); plain.synchronize()
    wS, wP = hip.download(dS, (n, total, 128), np.int16), hip.download(dP, (n, total, 128), np.int16)
    for p0, rows in host_rows.items():
        wS[:, p0:p0 + rows.shape[1]] = rows
    if in_place:
        wI = hip.download(dI2, (n, total, 128), np.int16)
        for p0, T in in_place:
            wS[:, p0:p0 + T] = wI[:, p0:p0 + T]
    if taps_on:
        tS, tP = subj.read_taps(), plain.read_taps()
        for k in tS:
            assert tS[k].tobytes() == tP[k].tobytes(), (seed, "tap", k)
    assert np.array_equal(wS, wP), "seed %d (n %d, %d shards, %d lanes): %d samples differ, first at %s" % (
        seed, n, shards, lanes, int((wS != wP).sum()), np.argwhere(wS != wP)[0].tolist())
    sS, sP = subj.read_status(), plain.read_status()
    for k in sS:
        assert sS[k].tobytes() == sP[k].tobytes(), (seed, k)
    hip.free_all(); subj.close(); plain.close()


@pytest.mark.parametrize("seed", _seeds())
def test_every_launch_form_equals_the_plain_one(gpu, ao, seed):
    run_sequence(gpu, seed)
