"""Pins the oracle's restatement of CMSIS-DSP `arm_biquad_cascade_df1_f32` / `_init_f32` against THE REFERENCE'S OWN BINARY.

The functions are the only third-party arithmetic on the hot path (call sites AudioSDR.cpp:77-78, 136-137, 285; prototypes
arm_math.h:1257-1262, 1360-1378) and the reference ships them only as Cortex-M4 objects inside
`ARM_MATH UPDATE/TeensyduinoArmMathUpdate/libarm_cortexM4lf_math.a`.  tests/thumb_emu.py reads the archive members and executes
their code; here:
  * a static census of the object: 25 vmul.f32, 20 vadd.f32, NO fused multiply-add of any kind (SURVEY.md 8(c) said so in prose);
  * the symbolic dataflow of one sample through the single-sample loop and of four through the 4x-unrolled loop:
    (((b0*x + b1*x1) + b2*x2) + a1*y1) + a2*y2, state written back as {x1, x2, y1, y2};
  * the object EXECUTED on random and adversarial inputs (every table of the reference, block sizes that take the unrolled loop,
    the tail loop and both, in place as AudioSDR.cpp calls it) == oracle.biquad_cascade, bit for bit, outputs and state;
  * `_init_f32` executed: the three instance fields and memset(pState, 0, 4 * numStages floats).
Those tests need /root/reference and skip without it (the GPU box); the committed vectors in tests/golden/cmsis_biquad_vectors.npz
(made by tests/golden/make_cmsis_vectors.py from the same emulated object) are checked against the oracle everywhere.
"""
import os

import numpy as np
import pytest

from oracle import asdr_oracle as ao
from tests import thumb_emu as T

ARCHIVE = "/root/reference/ARM_MATH UPDATE/TeensyduinoArmMathUpdate/libarm_cortexM4lf_math.a"
GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "cmsis_biquad_vectors.npz")
GOLDEN_CFFT = os.path.join(os.path.dirname(__file__), "golden", "cmsis_cfft128_vectors.npz")
needs_reference = pytest.mark.skipif(not os.path.exists(ARCHIVE), reason="the reference's CMSIS archive is not on this machine")

CODE, INST, COEF, STATE, SRC, DST, SP = 0x100, 0x1000, 0x1100, 0x1400, 0x2000, 0x6000, 0xF000


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def run_object(code, coefs, state, x, in_place=True, symbolic=False):
    """arm_biquad_cascade_df1_f32(&S, pSrc, pDst, blockSize) on the emulator -> (y, state after, cpu)."""
    coefs = np.asarray(coefs, dtype=np.float32).ravel()
    n_stages, n = coefs.size // 5, len(x)
    cpu = T.Cpu()
    cpu.symbolic = symbolic
    if symbolic:
        for st in range(n_stages):
            for k, nm in enumerate(("b0", "b1", "b2", "a1", "a2")): cpu.names[COEF + 20 * st + 4 * k] = "%s_%d" % (nm, st)
            for k, nm in enumerate(("x1", "x2", "y1", "y2")): cpu.names[STATE + 16 * st + 4 * k] = "%s_%d" % (nm, st)
        for k in range(n): cpu.names[SRC + 4 * k] = "in%d" % k
    cpu.load_code(code, CODE)
    cpu.wr32(INST, n_stages); cpu.wr32(INST + 4, STATE); cpu.wr32(INST + 8, COEF)     # arm_biquad_casd_df1_inst_f32, arm_math.h:1257-1262
    cpu.write_f32(COEF, coefs); cpu.write_f32(STATE, state); cpu.write_f32(SRC, x)
    dst = SRC if in_place else DST
    cpu.call(CODE, [INST, SRC, dst, n], SP)
    return cpu.read_f32(dst, n), cpu.read_f32(STATE, 4 * n_stages), cpu


def cases():
    """(name, coefs[n_stages*5], state[n_stages*4], x): the inputs of the committed vectors and of the live comparison."""
    rng = np.random.default_rng(20260410)
    out = []
    n_tables = 15
    for tbl in range(n_tables):                       # every biquad table the reference holds (AudioSDR.h:582-738)
        co = ao.biquad_table(tbl).ravel()
        for n in (128, 7):
            out.append(("table%d_n%d" % (tbl, n), co, (0.1 * rng.standard_normal(16)).astype(np.float32),
                        (0.5 * rng.standard_normal(n)).astype(np.float32)))
    co = ao.biquad_table(0).ravel()
    for n in (1, 2, 3, 4, 5, 8, 9, 127, 130):          # tail only | unrolled only | both
        out.append(("len%d" % n, co, (0.1 * rng.standard_normal(16)).astype(np.float32), rng.standard_normal(n).astype(np.float32)))
    for ns in (1, 2, 3):                               # fewer stages than the reference uses
        out.append(("stages%d" % ns, co[:5 * ns], (0.1 * rng.standard_normal(4 * ns)).astype(np.float32), rng.standard_normal(37).astype(np.float32)))
    # products whose exact sum needs more than 24 bits: a fused multiply-add would round differently from mul-then-add
    x = (1.0 + rng.integers(1, 1 << 23, 128) * 2.0 ** -23).astype(np.float32)
    coa = (rng.choice([-1.0, 1.0], 20) * (1.0 + rng.integers(1, 1 << 23, 20) * 2.0 ** -23) * 0.3).astype(np.float32)
    out.append(("fma_sensitive", coa, np.zeros(16, np.float32), x))
    # denormals (the FPv4-SP unit runs without flush-to-zero by default) and signed zeros
    tiny = (rng.standard_normal(64) * 1e-41).astype(np.float32)
    out.append(("denormal", co, (rng.standard_normal(16) * 1e-42).astype(np.float32), tiny))
    out.append(("zeros", co, np.zeros(16, np.float32), np.concatenate([np.zeros(5, np.float32), -np.zeros(5, np.float32)])))
    return out


@needs_reference
def test_object_holds_only_separate_multiplies_and_adds():
    code, rel = T.load_function(ARCHIVE, "arm_biquad_cascade_df1_f32.o", "arm_biquad_cascade_df1_f32")
    assert len(code) == 416 and rel == []             # self-contained: no calls, no literal pool
    c = T.vfp_census(code)
    assert c["vmul.f32"] == 25 and c["vadd.f32"] == 20, c    # (4x unrolled + tail) x (5 products, 4 sums)
    assert c["fused"] == 0 and c["vsub.f32"] == 0 and c["vnmul.f32"] == 0 and c["vdiv.f32"] == 0 and c["other"] == 0 and c["f64"] == 0, c


def _sum5(b0, x, b1, x1, b2, x2, a1, y1, a2, y2):
    """(((b0*x + b1*x1) + b2*x2) + a1*y1) + a2*y2 in the emulator's notation (operands of a commutative operation sorted)."""
    def op(a, b, o): a, b = sorted([a, b]); return "(%s%s%s)" % (a, o, b)
    acc = op(op(b0, x, "*"), op(b1, x1, "*"), "+")
    acc = op(acc, op(b2, x2, "*"), "+")
    acc = op(acc, op(a1, y1, "*"), "+")
    return op(acc, op(a2, y2, "*"), "+")


@needs_reference
def test_dataflow_association_and_state_order():
    code, _ = T.load_function(ARCHIVE, "arm_biquad_cascade_df1_f32.o", "arm_biquad_cascade_df1_f32")
    co = np.arange(1, 6, dtype=np.float32)
    c = ("b0_0", "b1_0", "b2_0", "a1_0", "a2_0")
    # one sample: the tail loop
    _, _, cpu = run_object(code, co, np.zeros(4, np.float32), np.ones(1, np.float32), in_place=False, symbolic=True)
    y0 = _sum5(c[0], "in0", c[1], "x1_0", c[2], "x2_0", c[3], "y1_0", c[4], "y2_0")
    assert cpu.mem_expr[DST] == y0
    assert [cpu.mem_expr[STATE + 4 * k] for k in range(4)] == ["in0", "x1_0", y0, "y1_0"]      # {x1, x2, y1, y2}
    # four samples: the unrolled loop; five: unrolled + tail
    for n in (4, 5):
        _, _, cpu = run_object(code, co, np.zeros(4, np.float32), np.ones(n, np.float32), in_place=False, symbolic=True)
        xs = ["x2_0", "x1_0"] + ["in%d" % k for k in range(n)]
        ys = ["y2_0", "y1_0"]
        for k in range(n):
            ys.append(_sum5(c[0], xs[k + 2], c[1], xs[k + 1], c[2], xs[k], c[3], ys[k + 1], c[4], ys[k]))
            assert cpu.mem_expr[DST + 4 * k] == ys[-1], (n, k)
        assert [cpu.mem_expr[STATE + 4 * k] for k in range(4)] == [xs[-1], xs[-2], ys[-1], ys[-2]]
    # two stages: stage 1 reads stage 0's OUTPUT (pDst), stage-major
    co2 = np.arange(1, 11, dtype=np.float32)
    _, _, cpu = run_object(code, co2, np.zeros(8, np.float32), np.ones(1, np.float32), in_place=False, symbolic=True)
    s0 = _sum5("b0_0", "in0", "b1_0", "x1_0", "b2_0", "x2_0", "a1_0", "y1_0", "a2_0", "y2_0")
    assert cpu.mem_expr[DST] == _sum5("b0_1", s0, "b1_1", "x1_1", "b2_1", "x2_1", "a1_1", "y1_1", "a2_1", "y2_1")


@needs_reference
def test_executed_object_equals_the_oracle():
    code, _ = T.load_function(ARCHIVE, "arm_biquad_cascade_df1_f32.o", "arm_biquad_cascade_df1_f32")
    for name, co, st, x in cases():
        for in_place in (True, False):
            y, st_after, cpu = run_object(code, co, st, x, in_place=in_place)
            yo, so = ao.biquad_cascade(co, st.copy(), x)
            assert np.array_equal(_bits(y), _bits(yo)), (name, in_place)
            assert np.array_equal(_bits(st_after), _bits(so)), (name, in_place)
            ns, n = len(co) // 5, len(x)
            assert cpu.trace.count("vmul.f32") == 5 * ns * n and cpu.trace.count("vadd.f32") == 4 * ns * n


@needs_reference
def test_init_object():
    code, rel = T.load_function(ARCHIVE, "arm_biquad_cascade_df1_init_f32.o", "arm_biquad_cascade_df1_init_f32")
    assert [(r[1], r[2]) for r in rel] == [(10, "memset")]      # R_ARM_THM_CALL memset: the one external call
    cpu = T.Cpu()
    cpu.load_code(code, CODE)
    cpu.mem[STATE - 16:STATE + 16 * 5] = b"\xAA" * (16 * 6)
    calls = []

    def memset(c):
        dst, val, n = c.r[0], c.r[1], c.r[2]
        calls.append((dst, val, n))
        c.mem[dst:dst + n] = bytes([val & 0xFF]) * n
    cpu.hooks[rel[0][0]] = memset
    cpu.call(CODE, [INST, 4, COEF, STATE], SP)       # (S, numStages, pCoeffs, pState), arm_math.h:1372-1378
    assert calls == [(STATE, 0, 4 * 4 * 4)]           # 4 state words per stage, zeroed
    assert (cpu.rd32(INST), cpu.rd32(INST + 4), cpu.rd32(INST + 8)) == (4, STATE, COEF)
    assert bytes(cpu.mem[STATE - 16:STATE]) == b"\xAA" * 16 and bytes(cpu.mem[STATE + 64:STATE + 80]) == b"\xAA" * 16


def test_committed_vectors_from_the_reference_binary_equal_the_oracle():
    """Runs everywhere (also on the GPU box): outputs of the reference's object, emulated here when the fixture was made."""
    g = np.load(GOLDEN)
    names = [str(n) for n in g["names"]]
    assert len(names) >= 40
    for i, name in enumerate(names):
        co, st, x = g["coefs_%d" % i], g["state_%d" % i], g["x_%d" % i]
        yo, so = ao.biquad_cascade(co, st.copy(), x)
        assert np.array_equal(_bits(yo), g["y_bits_%d" % i]), name
        assert np.array_equal(_bits(so), g["state_after_bits_%d" % i]), name


@needs_reference
def test_committed_vectors_are_what_the_object_computes():
    code, _ = T.load_function(ARCHIVE, "arm_biquad_cascade_df1_f32.o", "arm_biquad_cascade_df1_f32")
    g = np.load(GOLDEN)
    assert str(g["object_sha256"]) == __import__("hashlib").sha256(code).hexdigest()
    for i in (0, 7, len(g["names"]) - 3):
        y, st_after, _ = run_object(code, g["coefs_%d" % i], g["state_%d" % i], g["x_%d" % i])
        assert np.array_equal(_bits(y), g["y_bits_%d" % i]) and np.array_equal(_bits(st_after), g["state_after_bits_%d" % i])


@needs_reference
def test_cmplx_mag_squared_object():
    """`arm_cmplx_mag_squared_f32` (the line powers of AudioSDRpreProcessor.cpp:94) from the same archive: 10
    vmul.f32 + 5 vadd.f32, nothing fused (4x unrolled + tail), and executed it is re*re + im*im with every product and the sum
    rounded separately -- the association the front oracle (oracle/asdr_front_oracle.c:79-81) and asdr_pre_kernel use."""
    code, rel = T.load_function(ARCHIVE, "arm_cmplx_mag_squared_f32.o", "arm_cmplx_mag_squared_f32")
    c = T.vfp_census(code)
    assert rel == [] and c["vmul.f32"] == 10 and c["vadd.f32"] == 5 and c["fused"] == 0 and c["other"] == 0 and c["vsub.f32"] == 0, c
    rng = np.random.default_rng(5)
    for n in (128, 133, 3, 1):
        x = np.concatenate([rng.standard_normal(2 * n - 2) * 10.0 ** rng.integers(-20, 20), [1e-30, 3e-31]]).astype(np.float32)
        cpu = T.Cpu()
        cpu.load_code(code, CODE)
        cpu.write_f32(SRC, x)
        cpu.call(CODE, [SRC, DST, n], SP)                      # (pSrc, pDst, numSamples)
        got = cpu.read_f32(DST, n)
        with np.errstate(all="ignore"):
            re2 = (x[0::2] * x[0::2]).astype(np.float32); im2 = (x[1::2] * x[1::2]).astype(np.float32)
            want = (re2 + im2).astype(np.float32)
        assert np.array_equal(_bits(got), _bits(want)), n


# ------------------------------------------------------------------------------------------------------------------------------
# arm_cfft_f32, 128 points: the FFT of AudioSDRpreProcessor's image detector (AudioSDRpreProcessor.cpp:93)
def cfft_image():
    """The reference's objects linked in one emulator image: arm_cfft_f32 -> arm_cfft_radix8by2_f32 -> arm_radix8_butterfly_f32,
    arm_bitreversal_32, the instance arm_cfft_sR_f32_len128 and its tables (twiddleCoef_128, armBitRevIndexTable128)."""
    cpu = T.Cpu(1 << 17)
    img = T.Image(cpu)
    img.add(ARCHIVE, "arm_cfft_f32.o", [".text.arm_cfft_f32", ".text.arm_cfft_radix8by2_f32", ".text.arm_cfft_radix8by4_f32"])
    img.add(ARCHIVE, "arm_cfft_radix8_f32.o", [".text.arm_radix8_butterfly_f32"])
    img.add(ARCHIVE, "arm_bitreversal2.o", [".text"])
    img.add(ARCHIVE, "arm_common_tables.o", [".rodata.twiddleCoef_128", ".rodata.armBitRevIndexTable128"])
    img.add(ARCHIVE, "arm_const_structs.o", [".rodata.arm_cfft_sR_f32_len128"])
    img.link()
    return cpu, img


def run_cfft128(cpu, img, x):
    """arm_cfft_f32(&arm_cfft_sR_f32_len128, x, ifftFlag 0, bitReverseFlag 1) on the emulator, as the reference calls it."""
    buf, sp = 0x10000, 0x1F000
    cpu.write_f32(buf, x)
    cpu.call(img.sym["arm_cfft_f32"], [img.sym["arm_cfft_sR_f32_len128"], buf, 0, 1], sp, max_steps=5_000_000)
    return cpu.read_f32(buf, 256)


def cfft_cases():
    rng = np.random.default_rng(20260411)
    out = []
    imp = np.zeros(256, np.float32); imp[2 * 5] = 1.0
    out.append(("impulse", imp))
    t = np.arange(128)
    tone = np.empty(256, np.float32); tone[0::2] = np.cos(2 * np.pi * 20 * t / 128).astype(np.float32); tone[1::2] = np.sin(2 * np.pi * 20 * t / 128).astype(np.float32)
    out.append(("tone_line20", tone))
    for k in range(6):                                   # what the detector feeds it: int16 / 32767.0 (.cpp:89-90)
        out.append(("int16_block%d" % k, (rng.integers(-32768, 32768, 256).astype(np.float64) / 32767.0).astype(np.float32)))
    for k, scale in enumerate((1e-6, 1e-3, 1.0, 1e3)):
        out.append(("gauss_%g" % scale, (rng.standard_normal(256) * scale).astype(np.float32)))
    out.append(("denormal", (rng.standard_normal(256) * 1e-41).astype(np.float32)))
    out.append(("zeros_signed", np.concatenate([np.zeros(128, np.float32), -np.zeros(128, np.float32)])))
    return out


@needs_reference
def test_cfft128_objects_hold_no_fused_operations():
    for member, sec in (("arm_cfft_f32.o", ".text.arm_cfft_radix8by2_f32"), ("arm_cfft_radix8_f32.o", ".text.arm_radix8_butterfly_f32")):
        c = T.vfp_census(T.Elf32(T.ar_member(ARCHIVE, member)).section(sec))
        assert c["fused"] == 0 and c["f64"] == 0 and c["vdiv.f32"] == 0, (sec, c)
    c = T.vfp_census(T.Elf32(T.ar_member(ARCHIVE, "arm_cfft_radix8_f32.o")).section(".text.arm_radix8_butterfly_f32"))
    assert (c["vmul.f32"], c["vadd.f32"], c["vsub.f32"]) == (36, 59, 59), c      # both butterfly forms: 4 + 32 products, 118 sums


@needs_reference
def test_executed_cfft128_equals_the_oracle_and_the_float64_fft():
    cpu, img = cfft_image()
    s = img.sym["arm_cfft_sR_f32_len128"]
    assert (cpu.rd16(s), cpu.rd16(s + 12)) == (128, 208)                          # fftLen, bitRevLength (arm_const_structs)
    for name, x in cfft_cases():
        want = run_cfft128(cpu, img, x)
        assert np.array_equal(_bits(ao.fft128_interleaved(x)), _bits(want)), name
    # ... and it IS the discrete Fourier transform (float64 reference, 1e-6 of the largest line)
    x = cfft_cases()[3][1]
    y = run_cfft128(cpu, img, x)
    ref = np.fft.fft(x[0::2].astype(np.float64) + 1j * x[1::2].astype(np.float64))
    assert np.abs((y[0::2] + 1j * y[1::2]) - ref).max() < 1e-6 * np.abs(ref).max()
    # a trace census of one transform: 776 products, 2,244 sums, nothing else in the VFP unit
    cpu.trace.clear()
    run_cfft128(cpu, img, x)
    assert (cpu.trace.count("vmul.f32"), cpu.trace.count("vadd.f32"), cpu.trace.count("vsub.f32")) == (776, 1122, 1122)


@needs_reference
def test_twiddle_table_rule_gives_the_archives_table():
    """asdr_cfft128_tw (tools/extract_front_tables.py: cos / sin rounded to 9 decimals, then to float32) == twiddleCoef_128 of the
    reference's arm_common_tables.o, all 256 words; and the butterfly's constant C81 is the float32 of 0.70710678118."""
    import struct
    body = T.Elf32(T.ar_member(ARCHIVE, "arm_common_tables.o")).section(".rodata.twiddleCoef_128")
    ref = np.frombuffer(body, dtype=np.float32)
    k = np.arange(128)
    mine = np.empty(256, np.float32)
    mine[0::2] = [np.float32(round(float(np.cos(2 * np.pi * i / 128)), 9)) for i in k]
    mine[1::2] = [np.float32(round(float(np.sin(2 * np.pi * i / 128)), 9)) for i in k]
    assert np.array_equal(_bits(mine), _bits(ref))
    code = T.Elf32(T.ar_member(ARCHIVE, "arm_cfft_radix8_f32.o")).section(".text.arm_radix8_butterfly_f32")
    assert struct.pack("<f", np.float32(0.70710678118)) in code                   # the literal pool word 0x3f3504f3


def test_committed_cfft128_vectors_from_the_reference_binary_equal_the_oracle():
    """Runs everywhere (also on the GPU box): outputs of the reference's arm_cfft_f32 objects, emulated when the fixture was made."""
    g = np.load(GOLDEN_CFFT)
    names = [str(n) for n in g["names"]]
    assert len(names) >= 12
    for i, name in enumerate(names):
        assert np.array_equal(_bits(ao.fft128_interleaved(g["x_%d" % i])), g["y_bits_%d" % i]), name
