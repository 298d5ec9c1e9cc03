"""An INDEPENDENT restatement of the reference's control-plane arithmetic, written in numpy scalar arithmetic straight from
the reference text (not from the oracle, not from the product's host code):

    AudioSDR.cpp:439-457   agc_init            AudioSDR.cpp:459-480   agc_createLookupTable
    AudioSDR.cpp:551-566   setAGCattackTime / setAGCreleaseTime / setAGChangTime
    AudioSDR.cpp:187-222   setDemodMode tuning offsets          AudioSDR.h:164-168    IF centre and bandwidths
    AudioSDR.h:249-284     SAM PLL constants                    AudioSDR.h:483-491    log2_approx_f32
    AudioSDR.h:238-239     blanker average constants

The oracle (oracle/asdr_oracle.c) and the product's host control plane (audiosdr_amd/csrc/asdr_host.cpp) were written by one
author and share structure, so comparing those two with each other is a common-mode check.  Here both
are compared, bit for bit, with a third statement that shares no text with either: every C promotion (float literal vs double
literal, `x / 2.0`, `1.0 - a`, int * float) is spelled out with np.float32 / np.float64 scalars; the only shared ingredient is
the host libm (expf, frexpf, exp, log), which is what the reference itself calls.  Runs without a GPU (ASDR_NO_DEVICE)."""
import ctypes as C
import ctypes.util

import numpy as np
import pytest

from helpers import f32_bits

f32, f64 = np.float32, np.float64
_m = C.CDLL(ctypes.util.find_library("m") or "libm.so.6")
_m.expf.argtypes = [C.c_float]; _m.expf.restype = C.c_float
_m.frexpf.argtypes = [C.c_float, C.POINTER(C.c_int)]; _m.frexpf.restype = C.c_float
_m.exp.argtypes = [C.c_double]; _m.exp.restype = C.c_double
_m.log.argtypes = [C.c_double]; _m.log.restype = C.c_double
FS = f32(44100.0)                     # AUDIO_SAMPLE_RATE_EXACT on Teensy 4.x
PI = f64(3.1415926535897932384626433832795)   # Arduino.h PI (a double literal)


def bits(v):
    return int(np.float32(v).view(np.uint32))


def expf(x):
    return f32(_m.expf(float(f32(x))))    # the argument is converted to float at the call (float expf(float))


def log2_approx_f32(x):
    """.h:483-491: frexpf, Horner in float, `+ exponent` adds an int to a float."""
    e = C.c_int(0)
    m = f32(_m.frexpf(float(abs(f32(x))), C.byref(e)))
    r = f32(f32(1.23149591368684) * m) - f32(4.11852516267426)
    r = f32(f32(f32(r) * m) + f32(6.02197014179219))
    r = f32(f32(f32(r) * m) - f32(3.13396450166353))
    return f32(r + f32(e.value))


def agc_table(thr, slope, knee):
    """.cpp:459-480 with tableSize + 1 = 130 entries.  `2.3025`, `20.0`, `2.0`, `1.0`, `6.026`, `128.0` are double literals."""
    thr, slope, knee = f32(thr), f32(slope), f32(knee)
    lin_lo = expf(f64(2.3025) * (f64(thr) - f64(knee) / f64(2.0)) / f64(20.0))
    lin_hi = expf(f64(2.3025) * (f64(thr) + f64(knee) / f64(2.0)) / f64(20.0))
    out = np.zeros(130, dtype=np.float32)
    for i in range(130):
        inp = f32(f64(f32(i)) / f64(128.0))
        in_db = f32(f64(6.026) * f64(log2_approx_f32(inp)))
        if inp < lin_lo:
            out[i] = f32(1.0)
        elif inp > lin_hi:
            out_db = f32(thr + f32(f32(in_db - thr) * slope))                      # all float
            out[i] = expf(f64(2.3025) * f64(f32(out_db - in_db)) / f64(20.0))
        else:
            u = f64(f32(in_db - thr)) + f64(knee) / f64(2.0)                        # float difference, then + double
            out_db = f32(f64(in_db) + ((f64(slope) - f64(1.0)) * u * u) / (f64(2.0) * f64(knee)))
            out[i] = expf(f64(2.3025) * f64(f32(out_db - in_db)) / f64(20.0))
    return out


def time_constant(ms):
    """.cpp:448 / :553: exp(log(0.1) / (AUDIO_SAMPLE_RATE_EXACT*t / 1000.0)): float product, double divide, double exp, float store."""
    a = f32(_m.exp(_m.log(0.1) / float(f64(f32(FS * f32(ms))) / f64(1000.0))))
    b = f32(f64(1.0) - f64(a))
    return a, b


def hang_count_init(ms):
    """.cpp:447: AUDIO_SAMPLE_RATE_EXACT*(_agc_hangTime / 1000.0): the division is in double, then float * double."""
    return int(f64(FS) * (f64(f32(ms)) / f64(1000.0)))


def hang_count_set(ms):
    """.cpp:565: _agc_hangTime*AUDIO_SAMPLE_RATE_EXACT / 1000.0: float product first."""
    return int(f64(f32(f32(ms) * FS)) / f64(1000.0))


IFC, BW_SSB, BW_CW, BW_WSPR, BW_AM = f32(6890.0), f32(3000.0), f32(1000.0), f32(1000.0), f32(8500.0)   # .h:164-168


def tuning_offset(mode):
    """.cpp:187-222 (`/ 2.0` makes the expression double; the result is stored in a float member)."""
    if mode == 1 or mode == 6:
        return f32(f64(IFC) - f64(BW_SSB) / f64(2.0))
    if mode == 0:
        return f32(f64(IFC) + f64(BW_SSB) / f64(2.0))
    if mode == 3:
        return f32(f64(IFC) - f64(BW_CW) / f64(2.0))
    if mode == 2:
        return f32(f64(IFC) + f64(BW_CW) / f64(2.0))
    return IFC     # AM, SAM


def pll_constants():
    """.h:249-284: twoPI = 2.0*PI and halfPI = 0.5*PI are doubles rounded into const float32_t; wn, zeta, Ka float literals;
    `2*zeta` and `2*Ka` int * float -> float; b0/b1 = float * double."""
    two_pi, half_pi = f32(f64(2.0) * PI), f32(f64(0.5) * PI)
    alpha = f32(0.995); beta = f32(f64(1.0) - f64(alpha))
    f_conv = f32(FS / two_pi)
    lo, hi = f32(f64(IFC) - f64(1000.0)), f32(f64(IFC) + f64(1000.0))
    wn, zeta, ka = f32(0.07), f32(0.707), f32(1000.0)
    tau1 = f32(ka / f32(wn * wn))
    tau2 = f32(f32(f32(2) * zeta) / wn)
    g = f32(f32(f32(2) * ka) / tau1)
    b0 = f32(f64(g) * (f64(1.0) + f64(2.0) * f64(tau2)))
    b1 = f32(f64(g) * (f64(1.0) - f64(2.0) * f64(tau2)))
    nb_beta = f32(f64(1.0) - f64(f32(0.995)))                     # .h:238-239
    return np.array([b0, b1, f32(-1.0), alpha, beta, f_conv, lo, hi, two_pi, half_pi, f32(two_pi / FS), nb_beta], dtype=np.float32)


AGC_SETTINGS = [(-60.0, 0.1, 2.0), (-40.0, 0.3, 6.0), (-20.0, 0.5, 1.0), (-75.5, 0.05, 10.0), (-6.0, 1.0, 0.5), (-30.0, 0.25, 12.0)]


@pytest.mark.parametrize("thr,slope,knee", AGC_SETTINGS)
def test_agc_gain_table(A, ao, thr, slope, knee):
    want = agc_table(thr, slope, knee)
    b = A.AudioSDRBatch(2, device=-1)
    o = ao.OracleSDR()
    for x in (b, o):
        x.setAGCthreshold(thr); x.setAGCslope(slope); x.setAGCkneeWidth(knee)
    got_p = np.array([b.getAGClookup(i, 1) for i in range(129)], dtype=np.float32)
    got_o = np.array([o.getAGClookup(i) for i in range(129)], dtype=np.float32)
    assert np.array_equal(f32_bits(got_p), f32_bits(want[:129])), "product table differs from the independent restatement"
    assert np.array_equal(f32_bits(got_o), f32_bits(want[:129])), "oracle table differs from the independent restatement"
    assert want[0] == 1.0 and (np.diff(want[1:129]) <= 0).all()       # a compressor: gain never rises with level
    b.close()


@pytest.mark.parametrize("ms", [2.0, 5.0, 10.0, 100.0, 250.0, 500.0, 0.7, 1234.5])
def test_agc_time_constants_and_hang(A, ao, ms):
    al, be = time_constant(ms)
    b = A.AudioSDRBatch(1, device=-1)
    o = ao.OracleSDR()
    for x in (b, o):
        x.setAGCattackTime(ms); x.setAGCreleaseTime(ms); x.setAGChangTime(ms)
    for got in ((b.getAAGalphaAttack(0), b.getAGCbetaAttack(0), b.getAGCalphaRelease(0), b.getAGCbetaRelease(0)),
                (o.getAAGalphaAttack(), o.getAGCbetaAttack(), o.getAGCalphaRelease(), o.getAGCbetaRelease())):
        assert [bits(v) for v in got] == [bits(al), bits(be), bits(al), bits(be)]
    assert b.chain_constants(0)[0] == hang_count_set(ms) == o.chain_constants()[0]
    b.close()


def test_power_on_state(A, ao):
    """agc_init (.cpp:439-457): attack 5 ms, release 500 ms, hang 100 ms -> 4410 samples through the init formula."""
    b = A.AudioSDRBatch(1, device=-1)
    o = ao.OracleSDR()
    a5, b5 = time_constant(5.0)
    a500, b500 = time_constant(500.0)
    for got in ((b.getAAGalphaAttack(0), b.getAGCbetaAttack(0), b.getAGCalphaRelease(0), b.getAGCbetaRelease(0)),
                (o.getAAGalphaAttack(), o.getAGCbetaAttack(), o.getAGCalphaRelease(), o.getAGCbetaRelease())):
        assert [bits(v) for v in got] == [bits(v) for v in (a5, b5, a500, b500)]
    assert b.chain_constants(0)[0] == o.chain_constants()[0] == hang_count_init(100.0) == 4410
    want = pll_constants()
    assert np.array_equal(f32_bits(b.chain_constants(0)[1]), f32_bits(want)), (b.chain_constants(0)[1], want)
    assert np.array_equal(f32_bits(o.chain_constants()[1]), f32_bits(want)), (o.chain_constants()[1], want)
    assert abs(float(want[0]) - 0.4057) < 1e-3 and abs(float(want[1]) + 0.3861) < 1e-3
    b.close()


@pytest.mark.parametrize("mode", range(7))
def test_tuning_offsets(A, ao, mode):
    b = A.AudioSDRBatch(1, device=-1)
    o = ao.OracleSDR()
    want = tuning_offset(mode)
    assert bits(b.setDemodMode(mode)) == bits(want) == bits(o.setDemodMode(mode))
    assert bits(b.getTuningOffset(0)) == bits(want)
    b.close()
