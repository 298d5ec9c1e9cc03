"""The host-pointer entry point asdr_update() (include/asdr.h; the boundary of the reference's data path: host-resident audio blocks,
AudioSDR.cpp:46-47, 158-167) is overlapped in channel-range chunks -- H2D(k + 1) || kernels(k) || D2H(k - 1).  Chunking is by
channels, so it must equal the device-pointer path bit for bit: for ragged chunk sizes, multi-block calls, settings mixes whose
schedule is not in channel order (every kernel kind + remainders), pinned and pageable caller buffers."""
import os

import numpy as np
import pytest

from tests.helpers import Hip

pytestmark = pytest.mark.gpu


def isolated(fn):
    """Runs the test in a process of its own (pytest on this one test id, ASDR_TEST_ISOLATED=1).  The tests that switch asdr_host_autopin ON register
    ordinary numpy memory with the runtime (hipHostRegister) and release it again; in a long-lived process that went on to allocate, free and copy
    other numpy arrays, the runtime aborted in a LATER, unrelated copy (no message; 4-6 full-suite runs of 8, at the first test of the next file) --
    what the header's contract warns of, reached through the allocator's reuse of addresses.  The product keeps the feature opt-in; the suite keeps
    its own process clean of it."""
    import functools
    import subprocess
    import sys

    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        if os.environ.get("ASDR_TEST_ISOLATED") == "1":
            return fn(*args, **kwargs)
        root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
        r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                            os.path.join("tests", os.path.basename(__file__)) + "::" + fn.__name__],
                           cwd=root, env=dict(os.environ, ASDR_TEST_ISOLATED="1"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-3000:]
    return wrapper


def _mix(batch, n, kind):
    if kind == "usb":
        batch.setDemodMode(1); batch.enableAudioFilter()
    elif kind == "c4":
        for m in range(7):
            for c in range(m, n, 7):
                batch.setDemodMode(m, ch=c)
        batch.enableALSfilter(); batch.setNoiseBlankerThresholdDb(10.0)
    else:   # halves: consecutive channel ranges with different kernel kinds, a long ALS filter, a few odd channels
        batch.setDemodMode(1); batch.enableAudioFilter()
        for c in range(n // 2, n):
            batch.setDemodMode(5, ch=c)
        for c in range(n // 3, n // 3 + 50):
            batch.enableALSfilter(ch=c); batch.setALSfilterParams(100, 0.3, 5, ch=c)
        for c in (1, 77, n - 2):
            batch.setDemodMode(4, ch=c); batch.setMute(1, ch=c)


def _device_reference(gpu, n, kind, I, Q, calls):
    hip = Hip()
    b = gpu.AudioSDRBatch(n)
    _mix(b, n, kind)
    outs, pos = [], 0
    for T in calls:
        dI, dQ = hip.upload(I[:, pos:pos + T]), hip.upload(Q[:, pos:pos + T])
        dO = hip.malloc(n * T * 256)
        b.update_device(dI, dQ, dO, T)
        b.synchronize()
        outs.append(hip.download(dO, (n, T, 128), np.int16))
        pos += T
    st = b.read_status()
    hip.free_all(); b.close()
    return outs, st


@pytest.mark.parametrize("kind,n,chunks", [("usb", 4104, 0), ("usb", 4104, 5), ("c4", 5003, 3), ("c4", 5003, 16), ("halves", 3001, 7), ("halves", 3001, 2)])
def test_chunked_host_path_equals_the_device_path(gpu, kind, n, chunks):
    from audiosdr_amd.synth import make_iq
    calls = (1, 3, 1, 2)
    total = sum(calls)
    fc = 6890.0 - 600.0 + 25.0 * (np.arange(n) % 9)
    I, Q = make_iq(n, total, fc=fc, A=0.25, m=0.3, noise=0.02, impulse_every=1500)
    want, want_st = _device_reference(gpu, n, kind, I, Q, calls)
    for pinned in (False, True):
        b = gpu.AudioSDRBatch(n)
        _mix(b, n, kind)
        b.set_host_chunks(chunks)
        pos = 0
        was = gpu.binding.host_autopin(0 if not pinned else -1)   # (the staged path is what the pageable leg is about: no auto-pinning of temporaries that happen to recur)
        for k, T in enumerate(calls):
            if pinned:
                hI, hQ, hO = (gpu.host_alloc((n, T, 128)) for _ in range(3))
                hI[:] = I[:, pos:pos + T]; hQ[:] = Q[:, pos:pos + T]; hO[:] = 0x1111
                b.update_into(hI, hQ, hO)
                got = hO.copy()
                for a in (hI, hQ, hO):
                    gpu.host_free(a)
            else:
                got = b.update(I[:, pos:pos + T], Q[:, pos:pos + T])
            info = b.host_path_info()
            assert info["pinned"] == pinned
            if chunks:
                assert info["chunks"] == min(chunks, max(1, n // 8))
            assert np.array_equal(got, want[k]), "call %d (%d blocks), pinned %s: %d samples differ" % (k, T, pinned, int((got != want[k]).sum()))
            pos += T
        st = b.read_status()
        for key in st:
            assert st[key].tobytes() == want_st[key].tobytes(), key
        b.close()
        gpu.binding.host_autopin(was)


def _libc_mmap():
    import ctypes as C
    libc = C.CDLL(None, use_errno=True)
    libc.mmap.restype = C.c_void_p; libc.mmap.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_long]
    libc.munmap.restype = C.c_int; libc.munmap.argtypes = [C.c_void_p, C.c_size_t]
    return libc


@isolated
def test_recurring_pageable_buffers_are_pinned_from_the_second_call(gpu):
    """Ordinary (pageable) caller buffers that come back -- the reference's audio library reuses its blocks -- are registered in place the second
    time asdr_update sees them and DMA-copied from then on (include/asdr.h "Pageable buffers that RECUR"); a one-shot buffer stays on the staged
    path; asdr_host_autopin(0) switches it off; the cache is bounded and is emptied with the process' last batch."""
    from audiosdr_amd.synth import make_iq
    n, T = 2048, 2
    I, Q = make_iq(n, 3 * T, fc=6290.0, A=0.25, impulse_every=900)
    want, _ = _device_reference(gpu, n, "usb", I, Q, (T, T, T))
    gpu.binding.host_autopin_clear()
    gpu.binding.host_autopin(1)
    base = gpu.binding.host_autopin_info()
    b = gpu.AudioSDRBatch(n)
    _mix(b, n, "usb")
    hI, hQ, hO = (np.zeros((n, T, 128), np.int16) for _ in range(3))   # the application's own buffers, reused call after call
    for k in range(3):
        hI[:] = I[:, k * T:(k + 1) * T]; hQ[:] = Q[:, k * T:(k + 1) * T]; hO[:] = 0x1111
        b.update_into(hI, hQ, hO)
        assert b.host_path_info()["pinned"] == (k >= 1), "call %d" % k
        assert np.array_equal(hO, want[k]), "call %d: %d samples differ" % (k, int((hO != want[k]).sum()))
    info = gpu.binding.host_autopin_info()
    assert info["registered_now"] == 3 and info["registrations"] - base["registrations"] == 3 and info["refused"] == base["refused"], info
    # many other recurring ranges: the cache stays bounded (12 ranges), the oldest go
    small = gpu.AudioSDRBatch(64)
    bufs = [[np.zeros((64, 1, 128), np.int16) for _ in range(3)] for _ in range(8)]
    for rep in range(2):
        for tri in bufs:
            small.update_into(*tri)
    assert gpu.binding.host_autopin_info()["registered_now"] <= 12
    small.close()
    # off: a recurring buffer stays on the staged path
    gpu.binding.host_autopin(0)
    x = [np.zeros((n, T, 128), np.int16) for _ in range(3)]
    for k in range(3):
        b.update_into(*x)
        assert not b.host_path_info()["pinned"]
    gpu.binding.host_autopin(1)
    b.close()
    assert gpu.binding.host_autopin_info()["registered_now"] == 0   # (the last batch of the process took the registrations with it)
    gpu.binding.host_autopin(0)                                      # the default


@isolated
def test_autopin_eviction_never_takes_a_range_another_call_is_copying_through(gpu):
    """Two threads, each with its own batch and six recurring buffer triples: 36 ranges compete for the cache's 12 places, so every registration
    evicts -- but never a range that the OTHER thread's call is copying through at that moment (a call holds its ranges for its length).  Every
    result is the device path's."""
    import threading
    from audiosdr_amd.synth import make_iq
    n, T, reps, ntri = 512, 1, 3, 6   # (2 x 6 x 3 = 36 ranges: more than the 12 that stay registered, fewer than the 48 sightings the table remembers)
    I, Q = make_iq(n, reps * ntri, fc=6290.0, A=0.25)
    wants = []
    for t in range(2):
        w, _ = _device_reference(gpu, n, "usb", I, Q, (T,) * (reps * ntri))
        wants.append(w)
    gpu.binding.host_autopin_clear()
    gpu.binding.host_autopin(1)
    errs = []

    # (the buffers outlive the cache's registrations -- the contract of asdr_host_autopin: they are freed behind `keep.close()`, which empties the cache;
    # freed inside the workers, their addresses came back to later tests' arrays while still registered, and the process aborted in a copy)
    all_tris = [[[np.zeros((n, T, 128), np.int16) for _ in range(3)] for _ in range(ntri)] for _ in range(2)]

    def worker(t):
        try:
            b = gpu.AudioSDRBatch(n)
            _mix(b, n, "usb")
            tris = all_tris[t]
            k = 0
            for rep in range(reps):
                for tri in tris:
                    tri[0][:] = I[:, k:k + 1]; tri[1][:] = Q[:, k:k + 1]; tri[2][:] = 0x2222
                    b.update_into(*tri)
                    if not np.array_equal(tri[2], wants[t][k]):
                        errs.append("thread %d call %d: %d samples differ" % (t, k, int((tri[2] != wants[t][k]).sum())))
                    k += 1
            b.close()
        except Exception as e:   # noqa: BLE001
            errs.append("thread %d: %r" % (t, e))

    keep = gpu.AudioSDRBatch(8)   # (the process' last batch takes the registrations with it: not before both threads are done)
    th = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    info = gpu.binding.host_autopin_info()
    keep.close()
    gpu.binding.host_autopin_clear()
    gpu.binding.host_autopin(0)
    assert gpu.binding.host_autopin_info()["registered_now"] == 0
    del all_tris
    assert not errs, errs[:3]
    assert info["registered_now"] <= 12 and info["registrations"] >= 12, info


def test_a_freed_and_reallocated_buffer_is_read_through_its_new_pages(gpu):
    """The hazard of pinning for the caller: the application frees a buffer and gets the same address back for new data.  A range is mapped at a
    fixed address, used twice, unmapped, mapped AGAIN at the same address with other samples, used again: the call must see the new samples and
    produce the device path's result for them.  With the library's defaults no caller range is ever registered behind the caller's back
    (asdr_host_autopin is opt-in BECAUSE this sequence, run with it on, aborted the process in the DMA through the stale registration), so
    the recurring range stays on the staged path -- asserted -- and the third call is exact."""
    import ctypes as C
    import mmap
    from audiosdr_amd.synth import make_iq
    libc = _libc_mmap()
    n, T = 4096, 2
    nbytes = n * T * 128 * 2
    I1, Q1 = make_iq(n, 2 * T, fc=6290.0, A=0.25)
    I2, Q2 = make_iq(n, T, fc=6100.0, A=0.35, m=0.5)
    want1, _ = _device_reference(gpu, n, "usb", I1, Q1, (T, T))
    gpu.binding.host_autopin_clear()
    assert gpu.binding.host_autopin(-1) == 0 or os.environ.get("ASDR_HOST_AUTOPIN"), "auto-pinning must be off unless the application asks for it"
    gpu.binding.host_autopin(0)
    PROT_RW, MAP_PRIVATE_ANON, MAP_FIXED = 3, 0x22, 0x10
    size = 3 * nbytes
    addr = libc.mmap(None, size, PROT_RW, MAP_PRIVATE_ANON, -1, 0)
    assert addr not in (None, C.c_void_p(-1).value)
    def views(a):
        return [np.ctypeslib.as_array((C.c_int16 * (nbytes // 2)).from_address(a + i * nbytes)).reshape(n, T, 128) for i in range(3)]
    hI, hQ, hO = views(addr)
    b = gpu.AudioSDRBatch(n)
    _mix(b, n, "usb")
    for k in range(2):
        hI[:] = I1[:, k * T:(k + 1) * T]; hQ[:] = Q1[:, k * T:(k + 1) * T]
        b.update_into(hI, hQ, hO)
        assert np.array_equal(hO, want1[k])
    assert not b.host_path_info()["pinned"] and gpu.binding.host_autopin_info()["registered_now"] == 0   # nothing was registered behind the caller's back
    del hI, hQ, hO
    assert libc.munmap(addr, size) == 0                     # the application frees its buffers ...
    addr2 = libc.mmap(addr, size, PROT_RW, MAP_PRIVATE_ANON | MAP_FIXED, -1, 0)   # ... and gets the same address back
    assert addr2 == addr
    hI, hQ, hO = views(addr2)
    hI[:] = I2; hQ[:] = Q2; hO[:] = 0x2222
    # the reference result: the same state (two calls of the first signal), then the new samples, all through device rows
    want2, _ = _device_reference(gpu, n, "usb", np.concatenate([I1, I2], axis=1), np.concatenate([Q1, Q2], axis=1), (T, T, T))
    b.update_into(hI, hQ, hO)
    assert np.array_equal(hO, want2[2]), "%d samples differ: the call read or wrote pages of the old mapping" % int((hO != want2[2]).sum())
    b.close()
    gpu.binding.host_autopin_clear()
    del hI, hQ, hO
    libc.munmap(addr2, size)


def test_default_chunking_of_a_large_call_and_registered_memory(gpu):
    """A call large enough for the default plan to cut it (16,384 channels x 4 blocks = 16 MB per row set -> 8 chunks), on caller
    memory pinned in place with asdr_host_register."""
    from audiosdr_amd.synth import make_iq
    import ctypes as C
    n, T = 16384, 4
    bI, bQ = make_iq(512, T, fc=6290.0, A=0.25, impulse_every=900)
    I = np.ascontiguousarray(np.tile(bI, (n // 512, 1, 1))); Q = np.ascontiguousarray(np.tile(bQ, (n // 512, 1, 1)))
    out = np.zeros_like(I)
    L = gpu.load_library()
    for a in (I, Q, out):
        assert L.asdr_host_register(a.ctypes.data_as(C.c_void_p), a.nbytes) == 0, L.asdr_last_error()
    b = gpu.AudioSDRBatch(n)
    b.setDemodMode(1); b.enableAudioFilter()
    b.update_into(I, Q, out)
    info = b.host_path_info()
    assert info["pinned"] and info["chunks"] == 8, info
    for a in (I, Q, out):
        assert L.asdr_host_unregister(a.ctypes.data_as(C.c_void_p)) == 0
    # all 512-channel tiles carry the same input and the same settings: identical audio; one tile against the device path
    assert np.array_equal(out.reshape(n // 512, 512, T, 128), np.broadcast_to(out[:512], (n // 512, 512, T, 128)))
    want, _ = _device_reference(gpu, 512, "usb", bI, bQ, (T,))
    assert np.array_equal(out[:512], want[0])
    b.close()


def test_missing_input_guard_and_errors_on_the_host_path(gpu):
    b = gpu.AudioSDRBatch(16)
    L = gpu.load_library()
    out = np.full((16, 1, 128), 7, np.int16)
    import ctypes as C
    p = C.POINTER(C.c_int16)
    assert L.asdr_update(b._h, None, out.ctypes.data_as(p), out.ctypes.data_as(p), 1) == 0      # AudioSDR.cpp:48-56: nothing happens
    assert (out == 7).all()
    assert L.asdr_update(b._h, out.ctypes.data_as(p), out.ctypes.data_as(p), None, 1) != 0
    b.close()
