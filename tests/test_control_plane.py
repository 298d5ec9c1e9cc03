"""Control plane of the product (host C++ behind the C ABI) vs the oracle: every getter after scripted
setter sequences, bit-for-bit.  Runs WITHOUT a GPU through a control-plane-only batch (ASDR_NO_DEVICE);
no signal-path call is made."""
import re
import os

import numpy as np
import pytest

from helpers import f32_bits

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
F_GETTERS = ["getTuningOffset", "getBPFlower", "getBPFupper", "getAGCthreshold", "getAGCslope", "getAGCkneeWidth",
             "getAGCattack", "getAGCrelease", "getAAGalphaAttack", "getAGCbetaAttack", "getAGCalphaRelease",
             "getAGCbetaRelease", "getAGCstaticGain"]
I_GETTERS = ["getMute", "getAudioFilter", "ALSfilterIsEnabled", "ALSfilterIsNotch", "ALSfilterIsPeak",
             "ALSfilterIsAdaptive", "AGCisEnabled", "NoiseBlankerisEnabled", "getDemodMode"]


def check_all(batch, o, ch):
    for g in F_GETTERS:
        a, b = getattr(batch, g)(ch), getattr(o, g)()
        assert f32_bits(np.float32(a)) == f32_bits(np.float32(b)), g
    for g in I_GETTERS:
        assert int(getattr(batch, g)(ch)) == int(getattr(o, g)()), g
    for i in range(130):
        assert f32_bits(np.float32(batch.getAGClookup(i, ch))) == f32_bits(np.float32(o.getAGClookup(i))), "lookup %d" % i


def test_defaults_after_construction(A, ao):
    b = A.AudioSDRBatch(3, device=-1)
    o = ao.OracleSDR()
    for ch in range(3):
        check_all(b, o, ch)
    # Appendix A of SURVEY.md / AudioSDR.cpp:174-185, 439-457
    assert b.getDemodMode(0) == A.LSBmode and b.getTuningOffset(0) == 8390.0
    assert b.getMute(0) == 0 and b.AGCisEnabled(0) == 1 and b.NoiseBlankerisEnabled(0) == 1 and b.ALSfilterIsEnabled(0) == 0
    assert b.getAGClookup(0, 0) == 1.0 and b.getAudioFilter(0) == 0
    b.close()


def test_tuning_offsets_and_band_limits(A, ao):
    """AudioSDR.cpp:187-222, 259-273 (including the `+-` of :271 for WSPR)."""
    b = A.AudioSDRBatch(1, device=-1)
    want = {A.LSBmode: (8390.0, 5390.0, 8390.0), A.USBmode: (5390.0, 5390.0, 8390.0), A.CW_LSBmode: (7390.0, 6390.0, 7390.0),
            A.CW_USBmode: (6390.0, 6390.0, 7390.0), A.AMmode: (6890.0, 2640.0, 11140.0), A.SAMmode: (6890.0, 2640.0, 11140.0),
            A.WSPRmode: (5390.0, 6390.0, 6390.0)}
    for m, (off, lo, hi) in want.items():
        assert b.setDemodMode(m) == off
        assert (b.getTuningOffset(0), b.getBPFlower(0), b.getBPFupper(0)) == (off, lo, hi)
    # unknown mode: only _mode changes (AudioSDR.cpp:188); band limits fall through to 0.0
    b.setDemodMode(A.USBmode); b.setDemodMode(9)
    assert b.getDemodMode(0) == 9 and b.getTuningOffset(0) == 5390.0 and b.getBPFlower(0) == 0.0
    b.close()


SCRIPTS = [
    [("setAGCmode", (1,))], [("setAGCmode", (2,))], [("setAGCmode", (3,))], [("setAGCmode", (0,))],
    [("setAGCthreshold", (-40.0,)), ("setAGCslope", (0.25,)), ("setAGCkneeWidth", (6.0,))],
    [("setAGChangTime", (250.0,))],                       # also lands in getAGClookup(129): AudioSDR.h:219-220
    [("setAGChangTime", (250.0,)), ("setAGCslope", (0.2,))],   # ... until the table is rebuilt
    [("setAGCattackTime", (0.7,)), ("setAGCreleaseTime", (1234.5,)), ("setAGCstaticGain", (3.0,))],
    [("setInputGain", (11.0,))], [("setInputGain", (-1.0,))],
    [("setIQgainBalance", (1.02,)), ("setInputGain", (2.0,))],   # balance silently dropped (AudioSDR.cpp:241)
    [("setAudioFilter", (2,))], [("setAudioFilter", (10,))], [("enableAudioFilter", ()), ("setAudioFilter", (10,))],
    [("setAudioFilter", (77,))],
    [("enableALSfilter", ()), ("setALSfilterPeak", ()), ("setALSfilterStatic", ())],
    [("setALSfilterParams", (200, 0.1, 9.0))],
    [("setMute", (1,))], [("disableNoiseBlanker", ())], [("setNoiseBlankerThresholdDb", (10.0,))],
    [("setDemodMode", (6,)), ("init", ())],
]


@pytest.mark.parametrize("script", SCRIPTS)
def test_setter_scripts(A, ao, script):
    b = A.AudioSDRBatch(2, device=-1)
    o = ao.OracleSDR()
    for meth, args in script:
        getattr(b, meth)(*args)          # broadcast (ch = ALL)
        getattr(o, meth)(*args)
    check_all(b, o, 0)
    check_all(b, o, 1)
    b.close()


def test_per_channel_setters_do_not_leak(A, ao):
    b = A.AudioSDRBatch(4, device=-1)
    b.setDemodMode(A.CW_USBmode, ch=2)
    b.setAGCthreshold(-30.0, ch=1)
    o_def, o_cw, o_thr = ao.OracleSDR(), ao.OracleSDR(), ao.OracleSDR()
    o_cw.setDemodMode(ao.CW_USBmode); o_thr.setAGCthreshold(-30.0)
    check_all(b, o_def, 0); check_all(b, o_thr, 1); check_all(b, o_cw, 2); check_all(b, o_def, 3)
    b.close()


def test_no_device_batch_refuses_signal_path(A):
    b = A.AudioSDRBatch(2, device=-1)
    z = np.zeros((2, 1, 128), np.int16)
    with pytest.raises(A.AsdrError, match="needs a HIP device"):
        b.update(z, z)
    with pytest.raises(A.AsdrError):
        b.read_status()
    b.close()


def test_c_abi_exports_every_declared_symbol(A):
    """include/asdr.h is the boundary: every function it declares must be exported by libasdr_hip.so."""
    import ctypes
    with open(os.path.join(ROOT, "include", "asdr.h")) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(asdr_\w+)\s*\(", text)))
    assert len(names) > 60
    L = ctypes.CDLL(A.library_path())
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    from audiosdr_amd.binding import EXPORTS
    assert set(EXPORTS) <= set(names)
    assert b"gfx950" in A.load_library().asdr_version()


def test_product_does_not_link_the_oracle(A):
    """The product path must never route through oracle/: no import in the package, no symbol in the library."""
    import subprocess
    pk = os.path.join(ROOT, "audiosdr_amd")
    for dirpath, _, files in os.walk(pk):
        for fn in files:
            if fn.endswith((".py", ".cpp", ".hip", ".h")):
                with open(os.path.join(dirpath, fn), errors="replace") as f:
                    src = f.read()
                assert "asdr_oracle" not in src and "from oracle" not in src and "import oracle" not in src, fn
    syms = subprocess.run(["nm", "-D", A.library_path()], capture_output=True, text=True).stdout
    assert "ao_update" not in syms and "ao_create" not in syms


def test_worked_examples_of_the_reference_manual(A, ao):
    """DOC/AudioSDR.pdf, section 2 "System Description" (tuning): the IF is centred at +6890 Hz; to tune an AM station at
    7,150,000 Hz the rf oscillator is set to 7,150,000 - 6890 = 7,143,110 Hz; an LSB signal is aligned with the upper edge of
    the 3 kHz SSB filter, 6890 + (3000/2) = 8390 Hz, oscillator 7,150,000 - 8390 = 7,141,610 Hz.  setDemodMode() returns that
    mode-dependent offset (BareBonesWSPR.ino:102,116 subtracts it from the LO)."""
    b = A.AudioSDRBatch(1, device=-1)
    o = ao.OracleSDR()
    for sdr_set in (lambda m: b.setDemodMode(m), lambda m: o.setDemodMode(m)):
        assert 7150000 - sdr_set(A.AMmode) == 7143110
        assert 7150000 - sdr_set(A.LSBmode) == 7141610
        assert sdr_set(A.USBmode) == 6890 - 3000 / 2
    # "8th order, elliptic IIR band-pass filter with a bandwidth of 1, 3, or 8 kHz, for CW, SSB, or AM" (manual, IF pre-filters)
    for mode, bw in ((A.CW_USBmode, 1000.0), (A.USBmode, 3000.0), (A.AMmode, 8500.0)):
        b.setDemodMode(mode)
        assert b.getBPFupper(0) - b.getBPFlower(0) == bw
    b.close()


def test_capture_sink_needs_a_device(A):
    """The capture sink and the strided entry point are signal-path calls: a control-plane-only batch refuses them."""
    b = A.AudioSDRBatch(4, device=A.NO_DEVICE) if hasattr(A, "NO_DEVICE") else A.AudioSDRBatch(4, device=-1)
    with pytest.raises(A.AsdrError, match="HBM|device"):
        b.capture_open(8)
    with pytest.raises(A.AsdrError, match="device"):
        b.update_device_strided(16, 16, 16, 1, 1, 1)
    assert b.capture_position == 0 and b.capture_capacity == 0
    b.close()
