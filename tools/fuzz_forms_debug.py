"""Replay one seed of tests/test_gpu_fuzz_launch_forms.py with a synchronisation and a comparison after EVERY call (GPU box):
    python tools/fuzz_forms_debug.py <seed>
A seed that fails in the suite and passes here is an ordering bug between two launch forms (round 4, seed 155: a call the block
pipeline took had been classified as a lane call first and skipped its wait for the previous call on another stream)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import audiosdr_amd as gpu
from tests.helpers import Hip
import tests.test_gpu_fuzz_launch_forms as F
from audiosdr_amd.synth import make_iq
seed = int(sys.argv[1])
rng = np.random.default_rng(1000 + seed)
n = int(rng.choice([40, 700, 4100, 8200 + int(rng.integers(0, 9)), 16384 + 5]))
uniq = min(n, 64); total = int(rng.integers(14, 30))
fc = 6890.0 - 400.0 + 30.0 * (np.arange(uniq) % 7)
bI, bQ = make_iq(uniq, total, fc=fc, A=0.3, m=0.4, noise=0.02, impulse_every=int(rng.choice([0, 1500, 4000])), f2=fc + 700.0, a2=0.1)
reps = (n + uniq - 1) // uniq
I = np.ascontiguousarray(np.tile(bI, (reps, 1, 1))[:n]); Q = np.ascontiguousarray(np.tile(bQ, (reps, 1, 1))[:n])
shards = int(rng.integers(1, 4))
subj = gpu.AudioSDRBatch(n, devices=[0] * shards) if shards > 1 else gpu.AudioSDRBatch(n)
plain = gpu.AudioSDRBatch(n); plain.set_lanes(False); plain.set_stream_pipeline(False)
lanes = int(rng.choice([1, 2, 3, 4])); subj.set_lanes(lanes, 64)
ops = F._random_settings(rng, n, big_groups=bool(rng.integers(0, 2)))
print("n", n, "total", total, "shards", shards, "lanes", lanes)
for o in ops: print("  setting", o[0], o[1], (o[2][0], o[2][-1], len(o[2])))
F._apply(subj, ops); F._apply(plain, ops)
hip = Hip(); s1 = hip.stream()
dI, dQ = hip.upload(I), hip.upload(Q)
dS, dP = hip.malloc(n * total * 256), hip.malloc(n * total * 256)
pos, host_rows = 0, {}
while pos < total:
    T = int(min(total - pos, rng.choice([1, 1, 1, 2, 3, 9, 12]))); off = pos * 256
    form = str(rng.choice(["batch", "batch", "stream", "null", "host", "host_pinned"]))
    desc = form
    if form in ("host", "host_pinned"):
        ck = int(rng.choice([0, 1, 2, 3, 5])); subj.set_host_chunks(ck); desc += " chunks %d" % ck
        if form == "host_pinned":
            hI, hQ, hO = (gpu.host_alloc((n, T, 128)) for _ in range(3))
            hI[:] = I[:, pos:pos + T]; hQ[:] = Q[:, pos:pos + T]
            subj.update_into(hI, hQ, hO); host_rows[pos] = hO.copy()
            for a in (hI, hQ, hO): gpu.host_free(a)
        else:
            host_rows[pos] = subj.update(I[:, pos:pos + T], Q[:, pos:pos + T])
    elif shards > 1 and rng.random() < 0.5:
        desc += " per-shard"
        for g in range(shards):
            lo, hi = subj.shard_range(g); o2 = off + lo * total * 256
            subj.shard(g).update_device_strided(dI + o2, dQ + o2, dS + o2, T, total, total, gpu.STREAM_BATCH if form == "batch" else (s1 if form == "stream" else 0))
    else:
        st = gpu.STREAM_BATCH if (form == "batch" and shards == 1) else (s1 if form == "stream" else 0)
        desc += " whole st=%s" % ("BATCH" if st == gpu.STREAM_BATCH else st)
        subj.update_device_strided(dI + off, dQ + off, dS + off, T, total, total, st)
    for k in range(T):
        plain.update_device_strided(dI + off + k * 256, dQ + off + k * 256, dP + off + k * 256, 1, total, total, 0)
    subj.synchronize(); plain.synchronize()
    wS, wP = hip.download(dS, (n, total, 128), np.int16), hip.download(dP, (n, total, 128), np.int16)
    if pos in host_rows: wS[:, pos:pos + T] = host_rows[pos]
    bad = (wS[:, pos:pos + T] != wP[:, pos:pos + T])
    print("call at block", pos, "T", T, desc, "differ:", int(bad.sum()), "channels:", int(bad.any(axis=(1, 2)).sum()), "pipeline launches", subj.stream_pipeline_launches(), "lane calls", subj.lane_calls())
    pos += T
    if rng.random() < 0.35:
        c = int(rng.integers(0, n))
        meth, args = [("setOutputGain", (float(rng.uniform(0.2, 1.0)),)), ("setDemodMode", (int(rng.choice(F.MODES)),)),
                      ("setAGChangTime", (float(rng.choice([0.0, 50.0])),)), ("enableALSfilter", ()), ("disableALSfilter", ()),
                      ("setMute", (int(rng.integers(0, 2)),))][int(rng.integers(0, 6))]
        print("   setter", meth, args, "ch", c)
        for b in (subj, plain): getattr(b, meth)(*args, ch=c)
    if rng.random() < 0.15:
        subj.read_status()
