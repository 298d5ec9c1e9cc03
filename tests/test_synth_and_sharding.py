"""Synthetic input determinism, and the N>1 sharding path on CPU (gloo, world_size 2).  On CPU the per-rank
compute stand-in is the oracle (tests may use it); the property under test is the sharding/gather logic that
bench.py and multi-GPU callers rely on: shard(rank) results concatenate to the single-process result."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def test_make_iq_subrectangles_are_bit_identical():
    from audiosdr_amd.synth import make_iq
    I, Q = make_iq(6, 5, fc=6290.0, A=0.25, m=0.3, impulse_every=300, f2=7000.0, a2=0.1)
    I2, Q2 = make_iq(2, 2, fc=6290.0, A=0.25, m=0.3, impulse_every=300, f2=7000.0, a2=0.1, channel0=3, start_block=2)
    assert np.array_equal(I[3:5, 2:4], I2) and np.array_equal(Q[3:5, 2:4], Q2)
    assert I.dtype == np.int16 and I.shape == (6, 5, 128)
    assert not np.array_equal(I[0], I[1])          # per-channel noise seeds differ
    assert 0.25 * 32767 < int(np.abs(I).max()) < (0.25 * 1.3 + 0.1 + 0.6 + 0.02) * 32767


def test_shard_ranges_partition():
    from audiosdr_amd.sharding import shard_range
    for n in (1, 7, 8, 65536, 1000003):
        for w in (1, 2, 4, 8):
            r = [shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n and all(r[i][1] == r[i + 1][0] for i in range(w - 1))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1


def _worker(rank, world, port, n_ch, n_blk, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from audiosdr_amd.sharding import shard_range
    from audiosdr_amd.synth import make_iq
    from oracle import asdr_oracle as ao
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(n_ch, rank, world)
    I, Q = make_iq(hi - lo, n_blk, fc=6290.0, A=0.25, channel0=lo)
    out, _ = ao.run_channels(lambda s, c: (s.setDemodMode(ao.USBmode), s.enableAudioFilter()), I, Q)
    dist.barrier()
    # gather per-rank checksums + shapes (what a multi-GPU caller does with its shards)
    sums = [None] * world
    dist.all_gather_object(sums, (lo, hi, int(out.astype(np.int64).sum()), out.tobytes()))
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t, op=dist.ReduceOp.MAX)     # the max-over-ranks timing reduction of bench.py
    if rank == 0:
        q.put((sums, float(t.item())))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_sharding_equals_single_process():
    import torch.multiprocessing as mp
    from audiosdr_amd.synth import make_iq
    from oracle import asdr_oracle as ao
    ao.build()
    n_ch, n_blk, world = 5, 6, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_ch, n_blk, q)) for r in range(world)]
    for p in procs:
        p.start()
    sums, tmax = q.get(timeout=300)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    assert tmax == float(world)
    I, Q = make_iq(n_ch, n_blk, fc=6290.0, A=0.25)
    want, _ = ao.run_channels(lambda s, c: (s.setDemodMode(ao.USBmode), s.enableAudioFilter()), I, Q)
    got = np.concatenate([np.frombuffer(s[3], dtype=np.int16).reshape(s[1] - s[0], n_blk, 128) for s in sums])
    assert [s[0] for s in sums] == [0, 2] and sums[-1][1] == n_ch
    assert np.array_equal(got, want)
