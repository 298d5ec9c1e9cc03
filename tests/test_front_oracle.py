"""CPU tests of the blocks around the hot path: the oracle restatements of
AudioSDRpreProcessor / AudioIQgenerator / AudioGrabberComplex256 against behaviour the reference's source and
comments document, float64 cross-checks, and the device-less control plane of the C ABI (include/asdr_front.h)
against the oracle.  No GPU needed."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
FS = 44100.0


def tone_iq(n_blocks, f, amp=0.3, q_delay=0, noise=0.0, seed=1):
    """Complex tone exp(+j 2 pi f t) as int16 I/Q streams; q_delay > 0 delays Q by that many samples (the Teensy I2S
    bug the pre-processor exists for), q_delay < 0 delays I instead."""
    n = n_blocks * 128
    t = np.arange(-4, n)
    rng = np.random.default_rng(seed)
    i = amp * np.cos(2 * np.pi * f * t / FS) + noise * rng.standard_normal(t.size)
    q = amp * np.sin(2 * np.pi * f * t / FS) + noise * rng.standard_normal(t.size)
    I = np.trunc(32767 * i).astype(np.int16)
    Q = np.trunc(32767 * q).astype(np.int16)
    di, dq = (0, q_delay) if q_delay >= 0 else (-q_delay, 0)
    return I[4 - di:4 - di + n].copy(), Q[4 - dq:4 - dq + n].copy()


# ----------------------------------------------------------------------------------------------------------
# AudioSDRpreProcessor
# ----------------------------------------------------------------------------------------------------------
def test_pre_defaults_pass_through(ao):
    """AudioSDRpreProcessor.h:72-83: no correction, no swap, detector off -> blocks pass unchanged."""
    p = ao.OraclePreProcessor()
    I, Q = tone_iq(3, 5000.0)
    i, q = p.update(I, Q)
    assert np.array_equal(i, I) and np.array_equal(q, Q)
    assert p.getI2SerrorCompensation() == 0 and p.getAutoI2SerrorDetectionStatus() == 0


def test_pre_correction_plus_one_delays_i_across_blocks(ao):
    """.cpp:62-66: I is delayed one sample; sample 127 of a block becomes sample 0 of the next; the first ever is 0."""
    p = ao.OraclePreProcessor()
    p.setI2SerrorCompensation(1)
    I = (np.arange(384) + 1).astype(np.int16)
    Q = (-np.arange(384) - 1).astype(np.int16)
    i, q = p.update(I, Q)
    assert np.array_equal(i, np.concatenate([[0], I[:-1]]))
    assert np.array_equal(q, Q)
    assert p.state()["saved_sample"] == I[-1]


def test_pre_correction_minus_one_quirk(ao):
    """.cpp:67-72: Q is shifted one sample within the block but its sample 0 keeps its value, and the carried
    Q sample is written into I[0] -- reproduced, not fixed."""
    p = ao.OraclePreProcessor()
    p.setI2SerrorCompensation(-1)
    I = (np.arange(256) + 1).astype(np.int16)
    Q = (-np.arange(256) - 1).astype(np.int16)
    i, q = p.update(I, Q)
    for b in range(2):
        s = slice(128 * b, 128 * b + 128)
        assert np.array_equal(q[s][1:], Q[s][:-1])
        assert q[s][0] == Q[s][0]
        assert np.array_equal(i[s][1:], I[s][1:])
    assert i[0] == 0 and i[128] == Q[127]
    assert p.state()["saved_sample"] == Q[255]


def test_pre_other_correction_values_do_nothing_and_cancel_autodetect(ao):
    """.cpp:160-163 stores any int; only +1 and -1 act (.cpp:62,67); setting it cancels auto-detection."""
    p = ao.OraclePreProcessor()
    p.startAutoI2SerrorDetection()
    assert p.getAutoI2SerrorDetectionStatus() == 1
    p.setI2SerrorCompensation(2)
    assert p.getAutoI2SerrorDetectionStatus() == 0 and p.getI2SerrorCompensation() == 2
    I, Q = tone_iq(2, 3000.0)
    i, q = p.update(I, Q)
    assert np.array_equal(i, I) and np.array_equal(q, Q)
    p.stopAutoI2SerrorDetection()
    assert p.getI2SerrorCompensation() == 0          # .cpp:152 "revert to no compensation"


def test_pre_swap_after_correction(ao):
    """.cpp:127-133: the swap is applied last, to the corrected blocks."""
    a, b = ao.OraclePreProcessor(), ao.OraclePreProcessor()
    for p in (a, b):
        p.setI2SerrorCompensation(1)
    b.swapIQ(True)
    I, Q = tone_iq(2, 4000.0)
    ia, qa = a.update(I, Q)
    ib, qb = b.update(I, Q)
    assert np.array_equal(ia, qb) and np.array_equal(qa, ib)


def test_fft128_matches_float64_fft(ao):
    """ao_fft128 -- CMSIS arm_cfft_f32 restated (bit-exact against the reference's objects: tests/test_cmsis_object.py) -- vs numpy's float64 FFT."""
    rng = np.random.default_rng(7)
    for _ in range(5):
        x = (rng.standard_normal(128) + 1j * rng.standard_normal(128)).astype(np.complex64)
        X = ao.fft128(x)
        R = np.fft.fft(x.astype(np.complex128))
        assert np.abs(X - R).max() <= 4e-7 * np.abs(R).max() * 7      # ~log2(N) roundings of float32
    e = np.zeros(128, dtype=np.complex64); e[1] = 1.0                  # unit impulse at n=1 -> the twiddles themselves
    X = ao.fft128(e)
    k = np.arange(128)
    assert np.abs(X - np.exp(-2j * np.pi * k / 128)).max() < 1e-7
    assert X[0] == 1.0 and X[32] == -1j and X[64] == -1.0 and X[96] == 1j   # exact quarter turns
    bins = np.exp(2j * np.pi * 9 * k / 128).astype(np.complex64)       # a bin-centred tone lands in one line
    X = ao.fft128(bins)
    assert abs(X[9] - 128) < 1e-4 and np.abs(np.delete(X, 9)).max() < 2e-5


def test_pre_detector_measurements_on_a_clean_tone(ao):
    """.cpp:96-107: strongest line, average over lines 5..122, image ratio = P[line]/P[128-line]."""
    p = ao.OraclePreProcessor()
    p.startAutoI2SerrorDetection()
    f = 20 * FS / 128                                  # bin 20
    I, Q = tone_iq(1, f, amp=0.3)
    p.update(I, Q)
    st = p.state()
    P = p.power_spectrum()
    x = (I.astype(np.float64) + 1j * Q.astype(np.float64)) / 32767.0
    R = np.abs(np.fft.fft(x)) ** 2
    assert st["max_line"] == 20 and st["strong"] == 1
    assert np.allclose(P, R, rtol=2e-5, atol=1e-9 * R.max())
    assert st["max_power"] == P[20]
    assert np.isclose(st["avg_power"], R[5:123].sum() / 118.0, rtol=1e-5)
    assert st["ratio"] > 1e4                            # clean quadrature: the image line is tiny
    assert st["failure_count"] == 0 and st["success_count"] == 1 and st["correction"] == 0


def test_pre_detector_ignores_weak_spectra(ao):
    """.cpp:109: lines that do not clear 10x the average leave every counter alone."""
    p = ao.OraclePreProcessor()
    p.startAutoI2SerrorDetection()
    rng = np.random.default_rng(3)
    I = (rng.standard_normal(128 * 20) * 2000).astype(np.int16)
    Q = (rng.standard_normal(128 * 20) * 2000).astype(np.int16)
    p.update(I, Q)
    st = p.state()
    assert st["strong"] == 0 and st["success_count"] == 0 and st["failure_count"] == 0 and st["correction"] == 0
    z = np.zeros(128, dtype=np.int16)                   # all-zero block: no line wins, ratio is 0/buffer[128]
    p.update(z, z)
    st = p.state()
    assert st["max_line"] == 0 and st["max_power"] == 0.0 and st["strong"] == 0


def test_pre_detector_finds_a_one_sample_q_delay(ao):
    """The documented purpose (.cpp:55-61, 75-81): Q late by one sample gives a poor image ratio; after more than
    maxFailureCount (10) consecutive failures the correction steps 0 -> +1, which delays I and restores balance."""
    p = ao.OraclePreProcessor()
    p.startAutoI2SerrorDetection()
    I, Q = tone_iq(40, 6890.0, amp=0.3, q_delay=1)
    corr = []
    for b in range(40):
        p.update(I[128 * b:128 * b + 128], Q[128 * b:128 * b + 128])
        corr.append(p.getI2SerrorCompensation())
    assert corr[:10] == [0] * 10 and corr[10] == 1 and corr[-1] == 1     # the 11th failing block flips it
    st = p.state()
    assert st["failure_count"] == 0 and st["ratio"] > 100.0
    # success counter restarted at the flip: 1 for the flipping block (.cpp:116-118), then one per strong block
    assert st["success_count"] == 1 + (40 - 11)


def test_pre_detector_cycles_through_corrections(ao):
    """.cpp:113-114: 0 -> 1 -> -1 -> 0 ...; an I-late skew is not cured by +1 (it makes it two samples), so the
    detector moves on to -1."""
    p = ao.OraclePreProcessor()
    p.startAutoI2SerrorDetection()
    I, Q = tone_iq(60, 6890.0, amp=0.3, q_delay=-1)
    seen = []
    for b in range(60):
        p.update(I[128 * b:128 * b + 128], Q[128 * b:128 * b + 128])
        c = p.getI2SerrorCompensation()
        if not seen or seen[-1] != c:
            seen.append(c)
    assert seen[:3] == [0, 1, -1]
    assert p.getI2SerrorCompensation() == -1 and p.state()["ratio"] > 10.0


def test_pre_detector_switches_itself_off_after_1001_successes(ao):
    """.cpp:120-122: successCount > maxSuccessCount (1000) clears autoDetectFlag and keeps the correction."""
    p = ao.OraclePreProcessor()
    p.startAutoI2SerrorDetection()
    I, Q = tone_iq(8, 20 * FS / 128, amp=0.3)
    n = 0
    while p.getAutoI2SerrorDetectionStatus() and n < 2000:
        p.update(I[:128], Q[:128]); n += 1
    assert n == 1001 and p.state()["success_count"] == 1001 and p.getI2SerrorCompensation() == 0
    before = p.state()
    p.update(I[:128], Q[:128])
    assert p.state() == before                           # detector no longer runs


def test_unit_scale_division_is_exact(ao):
    """The GPU's reciprocal form of s/32767.0 equals true binary64 division for every int16 (after float rounding)."""
    assert ao.lib().ao_front_check_div32767() == 0


# ----------------------------------------------------------------------------------------------------------
# AudioIQgenerator
# ----------------------------------------------------------------------------------------------------------
def _full_hilbert(taps):
    """The 257-tap impulse response the folded loop of AudioIQgenerator.cpp:65-73 realises (float64)."""
    h = np.zeros(257)
    # output i uses +c[k]*x[i - (2k+1)] - c[k]*x[i - 257 + 2(k+1)]  (x index relative to the newest block start + i)
    for k in range(64):
        h[2 * k + 1] += taps[k]
        h[257 - 2 * (k + 1)] -= taps[k]
    return h


def test_iqgen_tables_differ_from_audiosdr_only_at_tap_41(ao):
    """AudioIQgenerator.h:88-106 has +0.01159615 where AudioSDR.h has -0.01159615."""
    a, b = ao.iqgen_hilbert_taps(), ao.hilbert_taps()
    d = np.nonzero(a != b)[0]
    assert list(d) == [41] and a[41] == np.float32(0.01159615) and b[41] == -a[41]


def test_iqgen_matches_float64_convolution(ao):
    """I = input delayed 128 samples, Q = 257-tap FIR of the input (.cpp:52-82), within 1 LSB of a float64 model."""
    g = ao.OracleIQgenerator()
    rng = np.random.default_rng(5)
    n = 128 * 6
    x = (8000 * np.sin(2 * np.pi * 1500.0 * np.arange(n) / FS) + 500 * rng.standard_normal(n)).astype(np.int16)
    I, Q = g.update(x)
    xs = x.astype(np.float64) / 32767.0
    h = _full_hilbert(ao.iqgen_hilbert_taps().astype(np.float64))
    qref = np.convolve(xs, h)[:n] * 32767.0
    iref = np.concatenate([np.zeros(128), xs[:-128]]) * 32767.0
    assert np.abs(I - np.trunc(iref)).max() <= 1
    assert np.abs(Q - np.trunc(qref)).max() <= 1
    # a mid-band tone comes out in quadrature: Q lags I by 90 degrees, equal amplitude (it is a Hilbert transformer)
    t = np.arange(128 * 8)
    x = np.round(10000 * np.cos(2 * np.pi * 5000.0 * t / FS)).astype(np.int16)
    I, Q = ao.OracleIQgenerator().update(x)
    z = (I[512:] + 1j * Q[512:]).astype(np.complex128)
    ph = np.angle(z[1:] * np.conj(z[:-1])).mean() * FS / (2 * np.pi)
    assert abs(abs(ph) - 5000.0) < 5.0 and abs(np.abs(z).std() / np.abs(z).mean()) < 0.05


def test_iqgen_gain_balance(ao):
    """.h:55-59: gainI = balance, gainQ = 1/balance (binary64 division stored to float); outputs scale accordingly."""
    g0, g1 = ao.OracleIQgenerator(), ao.OracleIQgenerator()
    g1.setGainBalance(1.25)
    x = np.round(6000 * np.sin(2 * np.pi * 3000.0 * np.arange(128 * 5) / FS)).astype(np.int16)
    I0, Q0 = g0.update(x)
    I1, Q1 = g1.update(x)
    assert np.abs(I1 - np.trunc(I0 * 1.25)).max() <= 2 and np.abs(Q1 - np.trunc(Q0 / 1.25)).max() <= 2
    # conversion saturates at int32 and keeps the low half (documented convention): full-scale x 4 wraps
    g2 = ao.OracleIQgenerator(); g2.setGainBalance(4.0)
    x = np.full(128 * 3, 32767, dtype=np.int16)
    I2, _ = g2.update(x)
    assert I2[-1] == np.int16((32767 * 4) & 0xFFFF if (32767 * 4) & 0x8000 == 0 else ((32767 * 4) & 0xFFFF) - 65536)


# ----------------------------------------------------------------------------------------------------------
# AudioGrabberComplex256
# ----------------------------------------------------------------------------------------------------------
def test_grabber_protocol(ao):
    """.cpp:50-90: two blocks make one 256-point complex buffer (re, im interleaved); newDataAvailable()/grab()."""
    g = ao.OracleGrabber()
    I = np.arange(128 * 5, dtype=np.int16)
    Q = -np.arange(128 * 5, dtype=np.int16)
    d = np.full(512, 77, dtype=np.int16)
    assert g.newDataAvailable() == 0
    g.grab(d)
    assert (d == 77).all()                               # nothing valid yet: destination untouched (.cpp:81)
    g.update(I[:128], Q[:128])
    assert g.newDataAvailable() == 0                     # half a buffer
    g.update(I[128:256], Q[128:256])
    assert g.newDataAvailable() == 1
    g.grab(d)
    assert np.array_equal(d[0::2], I[:256]) and np.array_equal(d[1::2], Q[:256])
    assert g.newDataAvailable() == 0
    g.update(I[256:384], Q[256:384])
    g.grab(d)                                            # still the previous complete pair
    assert np.array_equal(d[0::2], I[:256])
    g.update(I[384:512], Q[384:512])
    assert g.newDataAvailable() == 1
    g.grab(d)
    assert np.array_equal(d[0::2], I[256:512]) and np.array_equal(d[1::2], Q[256:512])


# ----------------------------------------------------------------------------------------------------------
# C ABI (include/asdr_front.h): exports and the device-less control plane against the oracle
# ----------------------------------------------------------------------------------------------------------
def test_front_abi_exports_every_declared_symbol(A):
    import ctypes
    with open(os.path.join(ROOT, "include", "asdr_front.h")) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(asdr_(?:pre|iqgen|grab)_\w+)\s*\(", text)))
    assert len(names) >= 30
    L = ctypes.CDLL(A.library_path())
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    assert set(A.FRONT_EXPORTS) == set(names)


def test_pre_control_plane_without_a_device(A, ao):
    """Setters/getters of the batched pre-processor against the oracle's, on a control-plane-only batch."""
    n = 5
    b = A.AudioSDRpreProcessorBatch(n, device=A.NO_DEVICE)
    orc = [ao.OraclePreProcessor() for _ in range(n)]
    script = [("startAutoI2SerrorDetection", (), None), ("setI2SerrorCompensation", (-1,), 1), ("swapIQ", (True,), 2),
              ("stopAutoI2SerrorDetection", (), 3), ("setI2SerrorCompensation", (7,), 4), ("swapIQ", (False,), None),
              ("swapIQ", (True,), 0)]
    for meth, args, ch in script:
        for c in range(n):
            if ch is None or ch == c:
                getattr(orc[c], meth)(*args)
        getattr(b, meth)(*args, **({} if ch is None else {"ch": ch}))
    st = b.read_state()
    for c in range(n):
        assert b.getI2SerrorCompensation(c) == orc[c].getI2SerrorCompensation()
        assert b.getAutoI2SerrorDetectionStatus(c) == orc[c].getAutoI2SerrorDetectionStatus()
        o = orc[c].state()
        for k in ("correction", "saved_sample", "failure_count", "success_count", "auto_detect", "swap"):
            assert int(st[k][c]) == o[k], (k, c)
    z = np.zeros((n, 1, 128), dtype=np.int16)
    with pytest.raises(A.AsdrError, match="needs a HIP device"):
        b.update(z, z)
    b.close()
    for cls in (A.AudioIQgeneratorBatch, A.AudioGrabberComplex256Batch):
        g = cls(3, device=A.NO_DEVICE)
        with pytest.raises(A.AsdrError, match="needs a HIP device"):
            g.update(*([z[:3]] * (1 if cls is A.AudioIQgeneratorBatch else 2)))
        g.close()
    with pytest.raises(A.AsdrError):
        A.AudioSDRpreProcessorBatch(0, device=A.NO_DEVICE)


def test_grabber_power_spectrum_oracle_matches_float64_fft(ao):
    """Panadapter spectrum (this project's 256-point float32 FFT on the grabber buffer / 32768) vs numpy's float64 FFT;
    a complex tone lands in its bin, a negative-frequency tone in bin 256 - k."""
    rng = np.random.default_rng(11)
    b = rng.integers(-20000, 20000, 512).astype(np.int16)
    P = ao.grab_power_spectrum(b)
    x = (b[0::2].astype(np.float64) + 1j * b[1::2]) / 32768.0
    R = np.abs(np.fft.fft(x)) ** 2
    assert np.abs(P - R).max() <= 2e-6 * R.max()
    n = np.arange(256)
    for k in (5, 200):
        z = 0.5 * np.exp(2j * np.pi * k * n / 256)
        buf = np.empty(512, dtype=np.int16)
        buf[0::2] = np.round(32767 * z.real); buf[1::2] = np.round(32767 * z.imag)
        P = ao.grab_power_spectrum(buf)
        assert int(np.argmax(P)) == k and P[k] > 1e4 * np.delete(P, k).max()
