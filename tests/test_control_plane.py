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


def test_incremental_control_plane(A):
    """A setter marks only the channels it touched: the flush before the next launch refills those parameter rows and rebuilds
    the wave schedule only when a touched channel's schedule key changed (asdr_host.cpp flush_host).  One per-channel
    setOutputGain on a 1,048,576-channel batch must cost well under 50 us of host time."""
    import time
    n = 1 << 20
    b = A.AudioSDRBatch(n, device=-1)
    st = b.control_plane_flush()                      # first flush: everything
    assert st["rows_refilled"] == n + 1 and st["schedule_rebuilt"] and st["waves_plain"] == n // 8 and st["waves_sam"] == 0
    st = b.control_plane_flush()                      # nothing changed
    assert st["rows_refilled"] == 0 and not st["schedule_rebuilt"]
    b.setOutputGain(0.7, ch=17)                       # not part of the schedule key
    t0 = time.perf_counter()
    st = b.control_plane_flush()
    dt = time.perf_counter() - t0
    assert st["rows_refilled"] == 1 and not st["schedule_rebuilt"]
    best = dt
    for i in range(20):
        b.setOutputGain(0.5 + i * 0.01, ch=17 + i)
        t0 = time.perf_counter(); b.control_plane_flush(); best = min(best, time.perf_counter() - t0)
    assert best < 50e-6, "flush after one per-channel setter took %.1f us" % (best * 1e6)
    b.setInputGain(2.0, ch=5); b.setInputGain(3.0, ch=5); b.setNoiseBlankerThreshold(2.0, ch=9)   # same row twice: listed once
    st = b.control_plane_flush()
    assert st["rows_refilled"] == 2 and not st["schedule_rebuilt"]
    b.setDemodMode(A.SAMmode, ch=123)                 # kernel instantiation of one channel changes: schedule rebuilt
    st = b.control_plane_flush()
    assert st["rows_refilled"] == 1 and st["schedule_rebuilt"]
    assert st["waves_sam"] == 1 and st["waves_plain"] == (n - 1 + 7) // 8 and st["waves_als"] == 0
    b.enableALSfilter(ch=123); b.enableALSfilter(ch=7)   # two more kinds (SAM + ALS, ALS): one channel each
    st = b.control_plane_flush()
    # the remainders of all key groups -- 6 plain channels and these two -- share ONE wave of the general (ALS) instantiation
    assert st["schedule_rebuilt"] and st["waves_sam"] == 0 and st["waves_als"] == 1 and st["waves_plain"] == n // 8 - 1
    b.setALSfilterParams(100, 0.5, 3, ch=123); b.setALSfilterParams(100, 0.5, 3, ch=7)   # a long filter: another kind again
    st = b.control_plane_flush()
    assert st["schedule_rebuilt"] and st["waves_als"] == 1
    b.setMute(1)                                      # broadcast: bulk refill, flags are part of the key
    st = b.control_plane_flush()
    assert st["rows_refilled"] == n + 1 and st["schedule_rebuilt"]
    b.close()


def test_agc_table_pool_is_shared_and_compacted(A):
    """Distinct (threshold, slope, knee) triples share one 130-entry table each (hash lookup, reference counts); tables no
    channel uses any more are dropped at a flush once they outnumber the live ones (a knob sweep must not leak)."""
    b = A.AudioSDRBatch(4096, device=-1)
    assert b.control_plane_flush()["agc_tables_alive"] == 1
    for i in range(200):                              # a UI knob sweep on all channels
        b.setAGCthreshold(-60.0 + 0.1 * i)
    st = b.control_plane_flush()
    assert st["agc_tables_alive"] == 2                # the swept setting + the padding channel's power-on table
    for c in range(64):                               # per-channel settings: one table per distinct triple
        b.setAGCslope(0.1 + 0.01 * (c % 8), ch=c)
    st = b.control_plane_flush()
    assert st["agc_tables_alive"] == 9                # 8 slopes (one of them the batch-wide setting) + the padding channel's
    v = [b.getAGClookup(100, ch=c) for c in range(16)]
    assert v[0] == v[8] and v[1] == v[9] and v[0] != v[1]
    b.close()


def test_schedule_layout_kinds_and_remainders(A):
    """asdr_schedule_layout after a control-plane flush: which kernel kind a channel's settings select (asdr_host.cpp kernel_kind:
    the ALS filter on the compact rows needs taps <= 64, delay >= 0 and delay + taps <= 65 and a mode other than SAM; SAM + such a
    filter is a kind of its own), whole waves per settings group, ONE sub-range for the remainders of all groups, and the launch
    form of the SAM channels (three launches from 512 SAM channels)."""
    n = 4096
    b = A.AudioSDRBatch(n, device=-1)
    b.control_plane_flush()
    lay = b.schedule_layout()
    assert lay == {"plain": n, "sam": 0, "als_long": 0, "als_compact": 0, "sam_als": 0, "remainders": 0, "remainder_kind": -1,
                   "sam_three_launches": False, "als_two_launches": False}
    # channels 0..99: ALS with the default 55 taps / delay 3 -> compact; 96 in whole waves, 4 remainders (+ 4 plain remainders: n - 100 = 8k + 4)
    for c in range(100):
        b.enableALSfilter(ch=c)
    b.control_plane_flush()
    lay = b.schedule_layout()
    assert lay["als_compact"] == 96 and lay["plain"] == (n - 100) // 8 * 8 and lay["als_long"] == 0
    assert lay["remainders"] == 8 and lay["remainder_kind"] == 2      # 4 ALS + 4 plain channels: one wave of the general (ALS) kernel
    # the limits of the compact layout, 8 channels each (their own settings groups differ only in the filter shape -> same group
    # key, so they stay in the ALS sub-ranges by kind)
    shapes = {(64, 1): "als_compact", (64, 2): "als_long", (65, 0): "als_long", (1, 64): "als_compact", (1, 65): "als_long",
              (0, 3): "als_compact", (128, 0): "als_long"}
    base = 200
    for i, (m_d, kind) in enumerate(shapes.items()):
        for c in range(base + 8 * i, base + 8 * i + 8):
            b.enableALSfilter(ch=c); b.setALSfilterParams(m_d[0], 0.5, float(m_d[1]), ch=c)
    b.control_plane_flush()
    lay2 = b.schedule_layout()
    n_compact = sum(8 for k in shapes.values() if k == "als_compact"); n_long = sum(8 for k in shapes.values() if k == "als_long")
    assert lay2["als_compact"] + lay2["als_long"] + lay2["plain"] + lay2["remainders"] >= n
    assert lay2["als_long"] == n_long                                   # 4 x 8 channels, whole waves
    assert lay2["als_compact"] == 96 + n_compact
    # SAM: 600 channels -> the three-launch form; 8 of them with a compact ALS filter -> kind "sam_als"; below 512 -> the fused kernel
    for c in range(1000, 1600):
        b.setDemodMode(A.SAMmode, ch=c)
    for c in range(1000, 1008):
        b.enableALSfilter(ch=c)
    b.control_plane_flush()
    lay3 = b.schedule_layout()
    assert lay3["sam_three_launches"] and lay3["sam"] == 592 and lay3["sam_als"] == 8
    for c in range(1100, 1600):
        b.setDemodMode(A.USBmode, ch=c)
    b.control_plane_flush()
    lay4 = b.schedule_layout()
    # 92 SAM channels left (1008..1099): the fused kernel's general waves, padded to 96 slots; the 8 SAM + ALS channels keep their
    # kind (its launch then uses the long-row ALS kernel, which carries the PLL)
    assert not lay4["sam_three_launches"] and lay4["sam"] == 96 and lay4["sam_als"] == 8
    b.close()
