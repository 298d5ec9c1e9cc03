"""An INDEPENDENT restatement of the reference's signal path, in numpy scalar arithmetic, against the C oracle.

Test infrastructure only.  `oracle/asdr_oracle.c` and the HIP kernels were written by one author from one reading of
`AudioSDR.cpp`; comparing those two with each other is a common-mode check.  The class below was written separately, straight
from the reference's text -- SRC/AudioSDRlib/AudioSDR.cpp:67-161 (update), :280-286 (audioFilter), :324-352 (ALSfilter), :404-436
(agcProcessor), :483-494 (agc_staticCompressor), :606-650 (impulse_noise_blanker), :688-749 (SAMdemod) and AudioSDR.h:358-526
(sin_f32, cos_f32, approx_atan2_f32, fast_sqrt_f32, freq_shifter) -- with every C++ promotion spelled out: `S(x)` rounds to
binary32, `D(x)` widens to binary64, a reference expression that mixes float operands with an unsuffixed literal is evaluated in
binary64 and rounded where the reference stores it.  It shares no code with the oracle; only DATA comes from there (the
biquad / Hilbert / sine literals, which tools/extract_tables.py checks against the reference, and the AGC gain table, which
tests/test_control_plane_independent.py restates separately).  It must agree with the oracle bit for bit: int16 audio of >= 8
blocks per case and the status getters.  (It does not change what pins the oracle -- the reference has no vectors and cannot be
built here -- it removes the risk that oracle and product share one misreading.)"""
import numpy as np
import pytest

S = np.float32
D = np.float64
N = 128
PI = D(3.1415926535897932384626433832795)      # Arduino.h
FS = S(44100.0)                                # AUDIO_SAMPLE_RATE_EXACT (Teensy 4.x), a float literal


def i16_wrap(v):
    """(int) of a double stored into an int16_t member: truncation toward zero, then the low 16 bits (ARM: saturating at int32)."""
    if v != v:
        t = 0
    elif v >= 2147483647.0:
        t = 2147483647
    elif v <= -2147483648.0:
        t = -2147483648
    else:
        t = int(v)
    t &= 0xFFFF
    return t - 65536 if t >= 32768 else t


class Biquad4:
    """arm_biquad_cascade_df1_f32 with 4 stages (CMSIS-DSP: acc = b0 x + b1 x1 + b2 x2 + a1 y1 + a2 y2, products and sums
    separately rounded in that order; state x1 x2 y1 y2 per stage; stage after stage over the block)."""

    def __init__(self, rows):
        self.set(rows)

    def set(self, rows):           # arm_biquad_cascade_df1_init_f32: new coefficients, state cleared
        self.c = [[S(v) for v in r] for r in np.asarray(rows, dtype=np.float32).reshape(4, 5)]
        self.s = [[S(0)] * 4 for _ in range(4)]

    def run(self, x):
        y = [S(v) for v in x]
        for st in range(4):
            b0, b1, b2, a1, a2 = self.c[st]
            x1, x2, y1, y2 = self.s[st]
            out = []
            for xn in y:
                acc = S(b0 * xn)
                acc = S(acc + S(b1 * x1))
                acc = S(acc + S(b2 * x2))
                acc = S(acc + S(a1 * y1))
                acc = S(acc + S(a2 * y2))
                x2, x1, y2, y1 = x1, xn, y1, acc
                out.append(acc)
            self.s[st] = [x1, x2, y1, y2]
            y = out
        return y


class RefSDR:
    """One AudioSDR instance (only what update() touches), AudioSDR.h:161-284 defaults and init() (.cpp:174-185)."""

    def __init__(self, ao):
        self.ao = ao
        self.sine = [S(v) for v in ao.sine_table()]
        self.hil = [S(v) for v in ao.hilbert_taps()]
        self._tables()
        self.twoPI = S(D(2.0) * PI)
        self.in_i = S(1.0); self.in_q = S(1.0); self.out_gain = S(0.5)
        self.mode = 0; self.muted = False
        self.audio = [S(0)] * N
        self.I = [S(0)] * N; self.Q = [S(0)] * N
        self.bufI = [S(0)] * (4 * N); self.bufQ = [S(0)] * (4 * N)
        self.phase_ssb = S(0); self.phase_am = S(0)
        # ALS
        self.M = 55; self.delay = 3; self.lam = S(0.5)
        self.als_in = [S(0)] * (2 * N); self.als_w = [S(0)] * N
        self.als_on = False; self.notch = True; self.adaptive = True
        # AGC (agc_init .cpp:439-457: the control-plane numbers come from the oracle's getters; their arithmetic is pinned elsewhere)
        self.carrier = S(0); self.agc_gain = S(0); self.absval = S(0); self.old_abs = S(0); self.hang_counter = 0
        self.agc_on = True; self.agc_active = True; self.static_gain = S(10.0)
        # blanker
        self.nbI = [S(0)] * (3 * N); self.nbQ = [S(0)] * (3 * N); self.mask = [S(1)] * (3 * N)
        self.nb_alpha = S(0.995); self.nb_beta = S(D(1.0) - D(self.nb_alpha)); self.nb_thr = S(1.2); self.nb_avg = S(10.0)
        self.nb_on = True; self.nb_detected = False
        # SAM PLL (AudioSDR.h:249-284)
        self.alpha_f = S(0.995); self.beta_f = S(D(1.0) - D(self.alpha_f)); self.f_conv = S(FS / self.twoPI)
        self.if_center = S(6890.0)
        self.lock_lo = S(D(self.if_center) - D(1000.0)); self.lock_hi = S(D(self.if_center) + D(1000.0))
        wn, zeta, Ka = S(0.07), S(0.707), S(1000.0)
        tau1 = S(Ka / S(wn * wn)); tau2 = S(S(S(2) * zeta) / wn)
        self.b0 = S(D(S(S(S(2) * Ka) / tau1)) * (D(1.0) + D(2.0) * D(tau2)))
        self.b1 = S(D(S(S(S(2) * Ka) / tau1)) * (D(1.0) - D(2.0) * D(tau2)))
        self.a1 = S(-1.0)
        self.yRe = S(0); self.yIm = S(0); self.prev_filt = S(0); self.d0 = S(0); self.d1 = S(0)
        self.phase_est = S(0); self.pll_freq = S(0); self.locked = False
        self.af_on = False
        self.af = Biquad4(self.tab_audio[6])                    # bw2700 (.cpp:175)
        self.ifI = Biquad4(self.tab_if["ssb"]); self.ifQ = Biquad4(self.tab_if["ssb"])
        self.imI = Biquad4(self.tab_img); self.imQ = Biquad4(self.tab_img)
        self.setDemodMode(0)

    def _tables(self):
        # pool order of audiosdr_amd/csrc/asdr_tables.h (ASDR_TBL_*): IF ssb, wspr, cw, am; AM image; audio filters 0..9 (AudioSDR.h:56-66 order)
        bt = self.ao.biquad_table
        self.tab_if = {"ssb": bt(0), "wspr": bt(1), "cw": bt(2), "am": bt(3)}
        self.tab_img = bt(4)
        self.tab_audio = [bt(5 + k) for k in range(10)]

    # ---- the control-plane calls the cases need (.cpp:187-222, 289-311, 384-398, 653-674) ----
    def setDemodMode(self, m):
        self.mode = m & 0xFFFF
        c, half = D(self.if_center), D(2.0)
        if self.mode == 1: self.shift = S(c - D(S(3000.0)) / half); t = "ssb"
        elif self.mode == 0: self.shift = S(c + D(S(3000.0)) / half); t = "ssb"
        elif self.mode == 6: self.shift = S(c - D(S(3000.0)) / half); t = "wspr"
        elif self.mode == 3: self.shift = S(c - D(S(1000.0)) / half); t = "cw"
        elif self.mode == 2: self.shift = S(c + D(S(1000.0)) / half); t = "cw"
        elif self.mode in (4, 5): self.shift = self.if_center; t = "am"
        else: return
        self.ifI.set(self.tab_if[t]); self.ifQ.set(self.tab_if[t])

    def enableAudioFilter(self): self.af_on = True
    def setAudioFilter(self, k):
        if k == 10: self.af_on = False
        elif 0 <= k <= 9: self.af.set(self.tab_audio[k])
    def enableALSfilter(self):
        self.als_on = True
        self.als_w = [S(0)] * N; self.als_in = [S(0)] * (2 * N)
    def setNoiseBlankerThresholdDb(self, db):
        self.nb_thr = S(np.power(S(10.0), S(D(S(db)) / D(20.0)), dtype=np.float32))   # powf
        self._nb_init()
    def disableNoiseBlanker(self): self.nb_on = False
    def _nb_init(self):
        self.nbI = [S(0)] * (3 * N); self.nbQ = [S(0)] * (3 * N); self.mask = [S(1)] * (3 * N)
    def take_agc_from(self, o):
        """alpha / beta / hang count / gain table: control-plane arithmetic, pinned by tests/test_control_plane_independent.py"""
        self.al_a = S(o.getAAGalphaAttack()); self.be_a = S(o.getAGCbetaAttack())
        self.al_r = S(o.getAGCalphaRelease()); self.be_r = S(o.getAGCbetaRelease())
        self.hang = int(o.getAGChangCount())
        self.table = [S(o.getAGClookup(i)) for i in range(130)]

    # ---- AudioSDR.h:358-377 ----
    def sin_f32(self, ph):
        ph = S(ph)
        if ph >= self.twoPI: ph = S(ph - self.twoPI)
        if D(ph) < D(0.0): ph = S(ph + self.twoPI)
        ip = int(D(ph) * D(65535.0) / D(self.twoPI)) & 0xFFFF            # (long) then uint16_t
        idx, delta = ip >> 8, ip & 0xFF
        v1, v2 = self.sine[idx], self.sine[idx + 1]
        return S(D(v1) + D(S(S(v2 - v1) * S(delta))) / D(256.0))

    def cos_f32(self, ph):
        return self.sin_f32(S(D(S(ph)) + PI / D(2.0)))

    def atan2(self, y, x):                                                 # AudioSDR.h:384-408
        halfPI = S(D(0.5) * PI)
        def at(z):
            n1, n2 = S(0.97239411), S(-0.19194795)
            return S(S(n1 + S(S(n2 * z) * z)) * z)
        if D(x) != D(0.0):
            if abs(x) > abs(y):
                z = S(y / x)
                if D(x) > D(0.0): return at(z)
                elif D(y) >= D(0.0): return S(D(at(z)) + PI)
                else: return S(D(at(z)) - PI)
            else:
                z = S(x / y)
                if D(y) > D(0.0): return S(S(-at(z)) + halfPI)
                else: return S(S(-at(z)) - halfPI)
        else:
            if D(y) > D(0.0): return halfPI
            elif D(y) < D(0.0): return S(-halfPI)
        return S(0.0)

    @staticmethod
    def fast_sqrt(x):                                                      # AudioSDR.h:434-446, n_iter = 1
        x = S(x)
        i = int(np.array([x], dtype=np.float32).view(np.uint32)[0])
        i = (i - (1 << 23)) & 0xFFFFFFFF
        i >>= 1
        i = (i + (1 << 29)) & 0xFFFFFFFF
        out = np.array([i], dtype=np.uint32).view(np.float32)[0]
        with np.errstate(all="ignore"):
            return S(D(0.5) * D(S(out + S(x / out))))

    def freq_shifter(self, f_shift, phase):                                # AudioSDR.h:508-526
        inc = S(S(f_shift) * S(self.twoPI / FS))
        ph = S(phase)
        for i in range(N):
            c, s = self.cos_f32(ph), self.sin_f32(ph)
            ti, tq = self.I[i], self.Q[i]
            self.I[i] = S(S(ti * c) - S(tq * s))
            self.Q[i] = S(S(tq * c) + S(ti * s))
            ph = S(ph + inc)
            if ph > self.twoPI: ph = S(ph - self.twoPI)
            elif D(ph) < D(0.0): ph = S(ph + self.twoPI)
        return ph

    # ---- .cpp:606-650 ----
    def blanker(self):
        dn = [S(0.933), S(0.750), S(0.500), S(0.250), S(0.067), S(0.0), S(0.0)]
        self.nb_detected = False
        for buf, new in ((self.nbI, self.I), (self.nbQ, self.Q)):
            buf[0:N] = buf[N:2 * N]; buf[N:2 * N] = buf[2 * N:3 * N]; buf[2 * N:3 * N] = list(new)
        self.mask[0:N] = self.mask[N:2 * N]; self.mask[N:2 * N] = self.mask[2 * N:3 * N]; self.mask[2 * N:3 * N] = [S(1)] * N
        for i in range(N - 50, 2 * N):
            mag = self.fast_sqrt(S(S(self.nbI[i] * self.nbI[i]) + S(self.nbQ[i] * self.nbQ[i])))
            if mag > S(self.nb_avg * self.nb_thr):
                for j in range(-10, 11):
                    self.mask[i + j] = S(0)
                self.nb_detected = True
            self.nb_avg = S(S(self.nb_alpha * self.nb_avg) + S(self.nb_beta * mag))
        for i in range(N, 2 * N):
            if self.mask[i] == 1.0 and self.mask[i - 1] == 0.0:           # (the else-if repeats this test: dead)
                for j in range(7):
                    self.mask[i - 7 + j] = dn[j]
        for i in range(N):
            self.I[i] = S(self.mask[i] * self.nbI[i]); self.Q[i] = S(self.mask[i] * self.nbQ[i])

    # ---- .cpp:688-749 ----
    def sam(self):
        for i in range(N):
            xr, xi = self.I[i], self.Q[i]
            dre = S(S(xr * self.yRe) + S(xi * self.yIm))
            dim = S(S(xi * self.yRe) - S(xr * self.yIm))
            err = self.atan2(dim, dre)
            self.d1 = self.d0
            self.d0 = S(err - S(self.a1 * self.d1))
            filt = S(S(self.b0 * self.d0) + S(self.b1 * self.d1))
            self.phase_est = S(D(self.phase_est) + D(S(filt + self.prev_filt)) / D(2.0))
            self.prev_filt = filt
            while D(self.phase_est) >= PI: self.phase_est = S(self.phase_est - self.twoPI)
            while D(self.phase_est) < -PI: self.phase_est = S(self.phase_est + self.twoPI)
            self.yRe = self.cos_f32(self.phase_est); self.yIm = self.sin_f32(self.phase_est)
            self.pll_freq = S(S(self.alpha_f * self.pll_freq) + S(self.beta_f * S(filt * self.f_conv)))
            self.locked = bool(self.pll_freq > self.lock_lo and self.pll_freq < self.lock_hi)
            if self.locked:
                ti, tq = self.I[i], self.Q[i]
                self.I[i] = S(S(ti * self.yRe) + S(tq * self.yIm))
                self.Q[i] = S(S(S(-ti) * self.yIm) + S(tq * self.yRe))

    # ---- .cpp:404-436, 483-494 ----
    def compress(self, u16):
        idx = u16 >> 8
        if idx > 127: idx = 127
        delta = S(D(S(u16 & 0xFF)) / D(256.0))
        return S(self.table[idx] + S(S(self.table[idx + 1] - self.table[idx]) * delta))

    def agc(self):
        for i in range(N):
            if self.mode == 4: self.absval = S(D(2.0) * D(self.carrier))
            else: self.absval = S(abs(D(self.audio[i])))
            if D(self.absval) > D(1.0): self.absval = S(1.0)
            if self.absval > self.old_abs:
                self.absval = S(S(self.al_a * self.old_abs) + S(self.be_a * self.absval))
                self.old_abs = self.absval
                self.hang_counter = self.hang
                self.agc_gain = self.compress(i16_wrap(D(self.absval) * D(32767.0)) & 0xFFFF)
            elif self.hang_counter > 0:
                self.hang_counter -= 1
            else:
                self.absval = S(S(self.al_r * self.old_abs) + S(self.be_r * self.absval))
                self.old_abs = self.absval
                self.agc_gain = self.compress(i16_wrap(D(self.absval) * D(32767.0)) & 0xFFFF)
            self.agc_active = bool(D(self.agc_gain) < D(0.99))
            out = S(S(self.agc_gain * self.static_gain) * self.audio[i])
            if D(out) > D(1.0): out = S(1.0)
            if D(out) < D(-1.0): out = S(-1.0)
            self.audio[i] = out

    # ---- .cpp:324-352 ----
    def als(self):
        count = 0
        self.als_in[0:N] = self.als_in[N:2 * N]; self.als_in[N:2 * N] = list(self.audio)
        for i in range(N, 2 * N):
            y = S(0.0)
            for j in range(self.M):
                y = S(y + S(self.als_w[j] * self.als_in[(i - self.delay) - j]))
            e = S(self.als_in[i] - y)
            if self.adaptive:
                if count == 0:
                    for j in range(self.M):
                        g = S(e * self.als_in[i - self.delay - j])
                        self.als_w[j] = S(self.als_w[j] + S(self.lam * g))
                count = (count + 1) % 4
            self.audio[i - N] = e if self.notch else y

    # ---- .cpp:39-168 ----
    def update(self, bi, bq):
        for i in range(N):
            self.I[i] = S(D(S(int(bi[i]))) / D(32767.0) * D(self.in_i))
            self.Q[i] = S(D(S(int(bq[i]))) / D(32767.0) * D(self.in_q))
        if self.nb_on: self.blanker()
        self.I = self.ifI.run(self.I); self.Q = self.ifQ.run(self.Q)
        m = self.mode
        if m in (1, 0, 3, 2, 6):
            self.phase_ssb = self.freq_shifter(S(-self.shift), self.phase_ssb)
            for buf, new in ((self.bufI, self.I), (self.bufQ, self.Q)):
                buf[0:N] = buf[N:2 * N]; buf[N:2 * N] = buf[2 * N:3 * N]; buf[2 * N:3 * N] = buf[3 * N:4 * N]; buf[3 * N:4 * N] = list(new)
            for i in range(N):
                acc = S(0.0)
                for k in range(257 // 4):
                    i1 = (3 * N + i) - (2 * k + 1); i2 = (3 * N + i) - 257 + 2 * (k + 1)
                    acc = S(acc + S(self.hil[k] * S(self.bufQ[i1] - self.bufQ[i2])))
                self.Q[i] = acc
                self.I[i] = self.bufI[3 * N + i - 128]
            for i in range(N):
                if m in (1, 3, 6): self.audio[i] = S(self.I[i] - self.Q[i])
                elif m in (0, 2): self.audio[i] = S(self.I[i] + self.Q[i])
        elif m in (4, 5):
            if m == 5:
                self.sam()
                self.audio = list(self.Q)
            if m == 4 or (m == 5 and not self.locked):
                self.phase_am = self.freq_shifter(S(-self.if_center), self.phase_am)
                self.I = self.imI.run(self.I); self.Q = self.imQ.run(self.Q)
                for i in range(N):
                    self.audio[i] = S(np.sqrt(S(S(self.I[i] * self.I[i]) + S(self.Q[i] * self.Q[i]))))
                    self.carrier = S(D(0.995) * D(self.carrier) + D(0.005) * D(abs(self.audio[i])))
        if self.af_on: self.audio = self.af.run(self.audio)
        if self.agc_on: self.agc()
        if self.als_on: self.als()
        if self.muted: return np.zeros(N, np.int16)
        return np.array([i16_wrap(D(S(self.out_gain * a)) * D(32767.0)) for a in self.audio], dtype=np.int16)


CASES = {
    # name: (setters applied to both, signal, blocks)
    "usb_blanker_impulses_audio_filter": ([("setDemodMode", 1), ("enableAudioFilter",)], dict(fc=6290.0, A=0.25, impulse_every=700), 9),
    "lsb_default": ([], dict(fc=8390.0 - 700, A=0.2), 8),
    "am_default": ([("setDemodMode", 4)], dict(fc=6890.0, A=0.3, m=0.5, fm=400.0), 8),
    "sam_locks": ([("setDemodMode", 5), ("setNoiseBlankerThresholdDb", 10.0), ("enableAudioFilter",), ("setAudioFilter", 0)],
                  dict(fc=6940.0, A=0.3, m=0.5, fm=400.0), 10),
    "sam_falls_back_to_the_envelope": ([("setDemodMode", 5), ("disableNoiseBlanker",)], dict(fc=9500.0, A=0.3, m=0.5), 8),
    "usb_als_notch": ([("setDemodMode", 1), ("setNoiseBlankerThresholdDb", 10.0), ("enableALSfilter",)],
                      dict(fc=6290.0, A=0.25, f2=7290.0, a2=0.125), 8),
    "cw_usb_nb_off": ([("setDemodMode", 3), ("disableNoiseBlanker",), ("enableAudioFilter",), ("setAudioFilter", 1)],
                      dict(fc=6390 + 700.0, A=0.2), 8),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_independent_restatement_agrees_with_the_oracle_bit_for_bit(ao, name):
    from audiosdr_amd.synth import make_iq
    setters, sig, n_blk = CASES[name]
    I, Q = make_iq(1, n_blk, **sig)
    o = ao.OracleSDR(pll_wrap_bound=False)
    r = RefSDR(ao)
    for s in setters:
        getattr(o, s[0])(*s[1:])
        getattr(r, s[0])(*s[1:])
    r.take_agc_from(o)
    with np.errstate(over="ignore", invalid="ignore"):
        for b in range(n_blk):
            want = o.update(I[0, b], Q[0, b])
            got = r.update(I[0, b], Q[0, b])
            assert np.array_equal(got, want), "%s block %d: first difference at sample %d" % (name, b, int(np.argmax(got != want)))
    assert bool(o.NoiseBlankerDetection()) == r.nb_detected
    assert bool(o.getSAMphaseLockStatus()) == r.locked
    assert np.float32(o.getSAMfrequency()).view(np.uint32) == np.float32(r.pll_freq).view(np.uint32)
    assert np.float32(o.getAMcarrierLevel()).view(np.uint32) == np.float32(r.carrier).view(np.uint32)
    assert bool(o.AGCisActive()) == r.agc_active
    if name == "sam_locks":
        assert r.locked
        assert r.carrier > 0           # ... after the envelope detector had run on the (partially rotated) early blocks (.cpp:132)
    if name == "sam_falls_back_to_the_envelope":
        assert not r.locked and r.carrier > 0     # a carrier 2.6 kHz off: outside the lock window (AudioSDR.h:254-255), envelope path in every block
