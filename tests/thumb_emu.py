"""A small Thumb-2 + VFPv4 (single precision) emulator -- TEST INFRASTRUCTURE ONLY.

Purpose: the one piece of third-party arithmetic on the hot path, CMSIS-DSP V1.4.5 `arm_biquad_cascade_df1_f32` /
`arm_biquad_cascade_df1_init_f32` (prototypes: reference `arm_math.h:1257-1262, 1360-1378`), exists in the reference only as
Cortex-M4 objects inside `ARM_MATH UPDATE/TeensyduinoArmMathUpdate/libarm_cortexM4lf_math.a`.  No ARM toolchain or disassembler is
in this image, so this module reads the archive member (ar + ELF32, pure Python), and EXECUTES its `.text.<function>` section
instruction by instruction with IEEE binary32 arithmetic (numpy float32: every VMUL / VADD rounds once, as the FPv4-SP unit does
with the default FPSCR: round to nearest, no flush to zero).  Running the reference's own binary on test inputs pins the oracle's
restatement of that function (tests/test_cmsis_object.py) and yields the committed vectors of tests/golden/cmsis_biquad_vectors.npz.

Only the instruction forms these two functions use are implemented; anything else raises `Unsupported` with the offending
halfwords, so a different archive build cannot be mis-executed silently.  Every executed instruction is also appended to
`Cpu.trace` as a mnemonic, which is what the census assertions of the test count.
"""
import struct

import numpy as np


class Unsupported(Exception):
    pass


# ---------------------------------------------------------------------------------------------------------------------------
# container formats
def ar_member(path, name):
    """Return the bytes of member `name` of a System V / GNU `ar` archive (long names through the `//` table)."""
    with open(path, "rb") as f:
        data = f.read()
    if data[:8] != b"!<arch>\n":
        raise ValueError("not an ar archive: %s" % path)
    pos, longnames = 8, b""
    while pos + 60 <= len(data):
        hdr = data[pos:pos + 60]
        ident = hdr[:16].decode("latin1").rstrip()
        size = int(hdr[48:58].decode("latin1").strip())
        body = data[pos + 60:pos + 60 + size]
        if ident == "//":
            longnames = body
        elif ident != "/" and not ident.startswith("/SYM"):
            if ident.startswith("/") and ident[1:].isdigit():
                off = int(ident[1:])
                end = longnames.index(b"\n", off)
                member = longnames[off:end].decode("latin1").rstrip("/")
            else:
                member = ident.rstrip("/")
            if member == name:
                return body
        pos += 60 + size + (size & 1)
    raise KeyError(name)


class Elf32:
    """Just enough of an ELF32 little-endian relocatable: sections by name, symbols, REL entries of a section."""

    def __init__(self, blob):
        if blob[:6] != b"\x7fELF\x01\x01":
            raise ValueError("not an ELF32 little-endian object")
        self.blob = blob
        (self.e_type, self.e_machine) = struct.unpack_from("<HH", blob, 16)
        e_shoff, = struct.unpack_from("<I", blob, 32)
        e_shentsize, e_shnum, e_shstrndx = struct.unpack_from("<HHH", blob, 46)
        self.sh = [struct.unpack_from("<IIIIIIIIII", blob, e_shoff + i * e_shentsize) for i in range(e_shnum)]
        strtab = self._body(self.sh[e_shstrndx])
        self.names = [self._cstr(strtab, s[0]) for s in self.sh]

    @staticmethod
    def _cstr(tab, off):
        return tab[off:tab.index(b"\0", off)].decode("latin1")

    def _body(self, s):
        return self.blob[s[4]:s[4] + s[5]] if s[1] != 8 else b""   # SHT_NOBITS has no bytes

    def section(self, name):
        return self._body(self.sh[self.names.index(name)])

    def symbols(self):
        i = [k for k, s in enumerate(self.sh) if s[1] == 2][0]      # SHT_SYMTAB
        tab, strs = self._body(self.sh[i]), self._body(self.sh[self.sh[i][6]])
        out = []
        for k in range(len(tab) // 16):
            st_name, st_value, st_size, st_info, st_other, st_shndx = struct.unpack_from("<IIIBBH", tab, k * 16)
            out.append((self._cstr(strs, st_name), st_value, st_size, st_info, st_shndx))
        return out

    def relocations(self, section_name):
        """[(offset, type, symbol name)] of `.rel<section_name>` (empty if there is none)."""
        rel = ".rel" + section_name
        if rel not in self.names:
            return []
        body, syms = self.section(rel), self.symbols()
        return [(o, info & 0xFF, syms[info >> 8][0]) for o, info in (struct.unpack_from("<II", body, k * 8) for k in range(len(body) // 8))]


# ---------------------------------------------------------------------------------------------------------------------------
# the processor
RETURN_ADDRESS = 0xFFFFFF00   # a branch here ends the run


class Cpu:
    """r0-r15, NZCV, s0-s31 (as raw binary32 patterns), a flat little-endian memory."""

    def __init__(self, mem_bytes=1 << 16):
        self.r = [0] * 16
        self.n = self.z = self.c = self.v = 0
        self.s = np.zeros(32, dtype=np.uint32)
        self.mem = bytearray(mem_bytes)
        self.trace = []
        self.it_conds = []     # conditions of the instructions still inside the current IT block
        self.hooks = {}        # code offset of a BL -> python callable(cpu) (an external call, e.g. memset)
        self.code_base = 0
        # symbolic shadow of the float dataflow: sx[i] = the expression register s<i> holds, built from the names given to memory
        # words in `names` (address -> str); float stores leave their expression in mem_expr.  Products and sums are written
        # with their operands sorted (both are commutative bit for bit), so only the ASSOCIATION shows.
        # Off by default: a recursive filter's expressions grow like Fibonacci numbers -- switch it on for a few samples only.
        self.symbolic = False
        self.sx = [None] * 32
        self.names = {}
        self.mem_expr = {}

    # memory
    def rd32(self, a): return struct.unpack_from("<I", self.mem, a)[0]
    def wr32(self, a, v): struct.pack_into("<I", self.mem, a, v & 0xFFFFFFFF)
    def rd16(self, a): return struct.unpack_from("<H", self.mem, a)[0]

    def load_code(self, code, at):
        self.mem[at:at + len(code)] = code
        self.code_base = at

    def write_f32(self, a, arr):
        arr = np.ascontiguousarray(arr, dtype=np.float32)
        self.mem[a:a + arr.nbytes] = arr.tobytes()

    def read_f32(self, a, n):
        return np.frombuffer(bytes(self.mem[a:a + 4 * n]), dtype=np.float32).copy()

    # float helpers: one IEEE binary32 operation, rounded once
    def _f(self, i): return self.s[i:i + 1].view(np.float32)[0]

    def _setf(self, i, val): self.s[i] = np.array([val], dtype=np.float32).view(np.uint32)[0]

    def _sym_load(self, i, a):
        if self.symbolic: self.sx[i] = self.mem_expr.get(a, self.names.get(a, "mem[0x%x]" % a))

    def _sym_store(self, a, i):
        if self.symbolic: self.mem_expr[a] = self.sx[i]

    def _sym_op(self, sd, sn, sm, op):
        if not self.symbolic: return
        a, b = sorted([str(self.sx[sn]), str(self.sx[sm])]) if op in "*+" else (str(self.sx[sn]), str(self.sx[sm]))
        self.sx[sd] = "(%s%s%s)" % (a, op, b)

    def _flags_sub(self, a, b):
        res = (a - b) & 0xFFFFFFFF
        self.n, self.z = res >> 31, int(res == 0)
        self.c = int(a >= b)
        self.v = int(((a ^ b) & (a ^ res)) >> 31 & 1)
        return res

    def _flags_add(self, a, b):
        full = a + b
        res = full & 0xFFFFFFFF
        self.n, self.z, self.c = res >> 31, int(res == 0), int(full > 0xFFFFFFFF)
        self.v = int((~(a ^ b) & (a ^ res)) >> 31 & 1)
        return res

    def _cond(self, cond):
        n, z, c, v = self.n, self.z, self.c, self.v
        return [z == 1, z == 0, c == 1, c == 0, n == 1, n == 0, v == 1, v == 0, c == 1 and z == 0, c == 0 or z == 1,
                n == v, n != v, z == 0 and n == v, z == 1 or n != v, True][cond]

    def _branch(self, target):
        self.r[15] = target & ~1 if target < RETURN_ADDRESS else RETURN_ADDRESS

    def call(self, entry, args, sp, max_steps=2_000_000):
        for i, a in enumerate(args):
            self.r[i] = a & 0xFFFFFFFF
        self.r[13], self.r[14], self.r[15] = sp, RETURN_ADDRESS | 1, entry
        with np.errstate(all="ignore"):
            for _ in range(max_steps):
                if self.r[15] == RETURN_ADDRESS:
                    return self.r[0]
                self.step()
        raise RuntimeError("step limit reached")

    # one instruction
    def step(self):
        pc = self.r[15]
        h = self.rd16(pc)
        wide = (h >> 11) in (0b11101, 0b11110, 0b11111)
        self.r[15] = pc + (4 if wide else 2)
        if self.it_conds:                      # inside an IT block: the instruction runs only if its condition holds
            cond = self.it_conds.pop(0)
            if not self._cond(cond):
                self.trace.append("it-skipped")
                return
        if wide:
            self._exec32(pc, h, self.rd16(pc + 2))
        else:
            self._exec16(pc, h)

    def _exec16(self, pc, h):
        r, t = self.r, self.trace
        if h == 0xBF00:
            t.append("nop")
        elif (h & 0xFF00) == 0xBF00:                         # IT{x{y{z}}} firstcond, mask
            first, mask = (h >> 4) & 15, h & 15
            n = 4 - ((mask & -mask).bit_length() - 1)        # instructions in the block
            conds = [first]
            for k in range(1, n):
                conds.append(first if ((mask >> (4 - k)) & 1) == (first & 1) else first ^ 1)
            self.it_conds = conds; t.append("it")
        elif (h & 0xFC00) == 0x1800:                         # ADDS / SUBS Rd, Rn, Rm
            a, b = r[(h >> 3) & 7], r[(h >> 6) & 7]
            r[h & 7] = self._flags_sub(a, b) if h & 0x0200 else self._flags_add(a, b); t.append("subs" if h & 0x0200 else "adds")
        elif (h & 0xFC00) == 0x1C00:                         # ADDS / SUBS Rd, Rn, #imm3
            a, b = r[(h >> 3) & 7], (h >> 6) & 7
            r[h & 7] = self._flags_sub(a, b) if h & 0x0200 else self._flags_add(a, b); t.append("subs" if h & 0x0200 else "adds")
        elif (h & 0xF800) == 0x2800:                         # CMP Rn, #imm8
            self._flags_sub(r[(h >> 8) & 7], h & 0xFF); t.append("cmp")
        elif (h & 0xFFC0) == 0x4280:                         # CMP Rn, Rm (low registers)
            self._flags_sub(r[h & 7], r[(h >> 3) & 7]); t.append("cmp")
        elif (h & 0xFF00) == 0x4500:                         # CMP Rn, Rm (high registers)
            self._flags_sub(r[(h & 7) | ((h >> 4) & 8)], r[(h >> 3) & 15]); t.append("cmp")
        elif (h & 0xF800) == 0x8800:                         # LDRH Rt, [Rn, #imm5*2]
            r[h & 7] = self.rd16(r[(h >> 3) & 7] + ((h >> 6) & 31) * 2); t.append("ldrh")
        elif (h & 0xF800) == 0x8000:                         # STRH Rt, [Rn, #imm5*2]
            struct.pack_into("<H", self.mem, r[(h >> 3) & 7] + ((h >> 6) & 31) * 2, r[h & 7] & 0xFFFF); t.append("strh")
        elif (h & 0xF800) == 0x9800:                         # LDR Rt, [sp, #imm8*4]
            r[(h >> 8) & 7] = self.rd32(r[13] + (h & 0xFF) * 4); t.append("ldr")
        elif (h & 0xF800) == 0x9000:                         # STR Rt, [sp, #imm8*4]
            self.wr32(r[13] + (h & 0xFF) * 4, r[(h >> 8) & 7]); t.append("str")
        elif (h & 0xFF00) == 0xB000:                         # ADD / SUB sp, #imm7*4
            d = (h & 0x7F) * 4
            r[13] = (r[13] - d if h & 0x80 else r[13] + d) & 0xFFFFFFFF; t.append("add-sp")
        elif (h & 0xF800) == 0x6800:                         # LDR Rt, [Rn, #imm5*4]
            r[h & 7] = self.rd32(r[(h >> 3) & 7] + ((h >> 6) & 31) * 4); t.append("ldr")
        elif (h & 0xF800) == 0x6000:                         # STR Rt, [Rn, #imm5*4]
            self.wr32(r[(h >> 3) & 7] + ((h >> 6) & 31) * 4, r[h & 7]); t.append("str")
        elif (h & 0xFF00) == 0x4600:                         # MOV Rd, Rm (high registers allowed)
            rd = (h & 7) | ((h >> 4) & 8)
            val = r[(h >> 3) & 15]
            if rd == 15: self._branch(val)
            else: r[rd] = val
            t.append("mov")
        elif (h & 0xFF00) == 0x4400:                         # ADD Rdn, Rm
            rd = (h & 7) | ((h >> 4) & 8)
            r[rd] = (r[rd] + r[(h >> 3) & 15]) & 0xFFFFFFFF; t.append("add")
        elif (h & 0xF800) == 0x3000:                         # ADDS Rdn, #imm8
            rd = (h >> 8) & 7; r[rd] = self._flags_add(r[rd], h & 0xFF); t.append("adds")
        elif (h & 0xF800) == 0x3800:                         # SUBS Rdn, #imm8
            rd = (h >> 8) & 7; r[rd] = self._flags_sub(r[rd], h & 0xFF); t.append("subs")
        elif (h & 0xF800) == 0x2000:                         # MOVS Rd, #imm8
            rd = (h >> 8) & 7; r[rd] = h & 0xFF; self.n, self.z = 0, int(r[rd] == 0); t.append("movs")
        elif (h & 0xFF87) == 0x4700:                         # BX Rm
            self._branch(r[(h >> 3) & 15]); t.append("bx")
        elif (h & 0xF800) == 0x0800:                         # LSRS Rd, Rm, #imm5 (imm5 = 0 means 32)
            sh = ((h >> 6) & 31) or 32; val = r[(h >> 3) & 7]
            self.c = (val >> (sh - 1)) & 1
            r[h & 7] = (val >> sh) if sh < 32 else 0
            self.n, self.z = 0, int(r[h & 7] == 0); t.append("lsrs")
        elif (h & 0xF800) == 0x0000 and h != 0:              # LSLS Rd, Rm, #imm5
            sh = (h >> 6) & 31; val = r[(h >> 3) & 7]
            if sh: self.c = (val >> (32 - sh)) & 1
            r[h & 7] = (val << sh) & 0xFFFFFFFF
            self.n, self.z = r[h & 7] >> 31, int(r[h & 7] == 0); t.append("lsls")
        elif (h & 0xF000) == 0xD000 and ((h >> 8) & 15) < 14:   # B<cond>
            off = h & 0xFF
            if off & 0x80: off -= 0x100
            if self._cond((h >> 8) & 15): self._branch(pc + 4 + off * 2)
            t.append("bcond")
        elif (h & 0xF800) == 0xE000:                         # B
            off = h & 0x7FF
            if off & 0x400: off -= 0x800
            self._branch(pc + 4 + off * 2); t.append("b")
        elif (h & 0xF500) == 0xB100:                         # CBZ / CBNZ
            off = (((h >> 9) & 1) << 6) | (((h >> 3) & 31) << 1)
            nz = (h >> 11) & 1
            if (r[h & 7] != 0) == bool(nz): self._branch(pc + 4 + off)
            t.append("cbnz" if nz else "cbz")
        elif (h & 0xFE00) == 0xB400:                         # PUSH {rlist[, lr]}
            regs = [i for i in range(8) if h >> i & 1] + ([14] if h >> 8 & 1 else [])
            r[13] -= 4 * len(regs)
            for k, i in enumerate(regs): self.wr32(r[13] + 4 * k, r[i])
            t.append("push")
        elif (h & 0xFE00) == 0xBC00:                         # POP {rlist[, pc]}
            regs = [i for i in range(8) if h >> i & 1] + ([15] if h >> 8 & 1 else [])
            for k, i in enumerate(regs):
                val = self.rd32(r[13] + 4 * k)
                if i == 15: self._branch(val)
                else: r[i] = val
            r[13] += 4 * len(regs); t.append("pop")
        else:
            raise Unsupported("16-bit %04x at +0x%x" % (h, pc - self.code_base))

    def _thumb_imm(self, i, imm3, imm8):
        """ThumbExpandImm for the plain cases (no rotation) -- all these functions use."""
        imm12 = (i << 11) | (imm3 << 8) | imm8
        if imm12 >> 10 == 0:
            mode = (imm12 >> 8) & 3
            b = imm12 & 0xFF
            return [b, b | b << 16, b << 8 | b << 24, b | b << 8 | b << 16 | b << 24][mode]
        rot = imm12 >> 7
        val = 0x80 | (imm12 & 0x7F)
        return ((val >> rot) | (val << (32 - rot))) & 0xFFFFFFFF

    def _exec32(self, pc, h1, h2):
        r, t = self.r, self.trace
        w = (h1 << 16) | h2
        # ---- VFP / coprocessor 10 (single precision)
        if (h1 & 0xEF00) == 0xEE00 and (h2 & 0x0F10) == 0x0A00:    # data processing, sz = 0
            opc1 = (h1 >> 4) & 0xB                                   # P.QR with the D bit masked out
            D, N, M, op = (h1 >> 6) & 1, (h2 >> 7) & 1, (h2 >> 5) & 1, (h2 >> 6) & 1
            sd, sn, sm = ((h2 >> 12) & 15) << 1 | D, (h1 & 15) << 1 | N, (h2 & 15) << 1 | M
            a, b = self._f(sn), self._f(sm)
            if opc1 == 0x2 and op == 0: self._setf(sd, a * b); self._sym_op(sd, sn, sm, "*"); t.append("vmul.f32")
            elif opc1 == 0x3 and op == 0: self._setf(sd, a + b); self._sym_op(sd, sn, sm, "+"); t.append("vadd.f32")
            elif opc1 == 0x3 and op == 1: self._setf(sd, a - b); self._sym_op(sd, sn, sm, "-"); t.append("vsub.f32")
            elif opc1 == 0xB and (h1 & 15) == 0 and (h2 >> 6) & 3 == 1: self.s[sd] = self.s[sm]; self.sx[sd] = self.sx[sm]; t.append("vmov.f32")
            elif opc1 == 0xB and (h1 & 15) == 1 and (h2 >> 6) & 3 == 1: self.s[sd] = self.s[sm] ^ 0x80000000; self.sx[sd] = "(-%s)" % self.sx[sm]; t.append("vneg.f32")
            else:
                fused = {0x0: "vmla/vmls", 0x1: "vnmla/vnmls", 0x9: "vfnma/vfnms", 0xA: "vfma/vfms", 0x8: "vdiv"}.get(opc1, "vfp-op")
                t.append(fused)
                raise Unsupported("VFP data-processing %s %08x at +0x%x" % (fused, w, pc - self.code_base))
        elif (h1 & 0xFF30) == 0xED10 and (h2 & 0x0F00) == 0x0A00:   # VLDR Sd, [Rn, #+-imm8*4]
            sd = ((h2 >> 12) & 15) << 1 | (h1 >> 6) & 1
            off = (h2 & 0xFF) * 4
            base = ((pc + 4) & ~3) if (h1 & 15) == 15 else r[h1 & 15]     # (literal pool: Align(PC, 4))
            addr = base + (off if h1 >> 7 & 1 else -off)
            self.s[sd] = self.rd32(addr); self._sym_load(sd, addr); t.append("vldr")
        elif (h1 & 0xFF30) == 0xED00 and (h2 & 0x0F00) == 0x0A00:   # VSTR Sd, [Rn, #+-imm8*4]
            sd = ((h2 >> 12) & 15) << 1 | (h1 >> 6) & 1
            off = (h2 & 0xFF) * 4
            addr = r[h1 & 15] + (off if h1 >> 7 & 1 else -off)
            self.wr32(addr, int(self.s[sd])); self._sym_store(addr, sd); t.append("vstr")
        elif (h1 & 0xFE00) == 0xEC00 and (h2 & 0x0E00) == 0x0A00:   # VLDM / VSTM / VPUSH / VPOP (IA with writeback, DB with writeback)
            P, U, D, W, L = (h1 >> 8) & 1, (h1 >> 7) & 1, (h1 >> 6) & 1, (h1 >> 5) & 1, (h1 >> 4) & 1
            dbl = (h2 >> 8) & 1
            rn, vd, imm8 = h1 & 15, (h2 >> 12) & 15, h2 & 0xFF
            first = (D << 4 | vd) * 2 if dbl else (vd << 1 | D)     # index into s[]
            words = imm8                                             # registers (single) or 2 x registers (double)
            if not W or P == U: raise Unsupported("VLDM/VSTM form %08x" % w)
            base = r[rn] - 4 * words if P else r[rn]
            for k in range(words):
                if L: self.s[first + k] = self.rd32(base + 4 * k); self._sym_load(first + k, base + 4 * k)
                else: self.wr32(base + 4 * k, int(self.s[first + k])); self._sym_store(base + 4 * k, first + k)
            r[rn] = base if P else r[rn] + 4 * words
            t.append(("vldm" if L else "vstm") + (".64" if dbl else ".32"))
        # ---- integer, 32-bit encodings
        elif (h1 & 0xFFD0) == 0xE890 or (h1 & 0xFFD0) == 0xE900:   # LDMIA / STMDB with writeback on sp (POP.W / PUSH.W)
            load = (h1 >> 4) & 1
            regs = [i for i in range(16) if h2 >> i & 1]
            rn = h1 & 15
            if not (h1 >> 5) & 1: raise Unsupported("LDM/STM without writeback %08x" % w)
            if load:
                for k, i in enumerate(regs):
                    val = self.rd32(r[rn] + 4 * k)
                    if i == 15: self._branch(val)
                    else: r[i] = val
                r[rn] += 4 * len(regs); t.append("pop.w")
            else:
                r[rn] -= 4 * len(regs)
                for k, i in enumerate(regs): self.wr32(r[rn] + 4 * k, r[i])
                t.append("push.w")
        elif (h1 & 0xFFEF) == 0xEA4F:                                # MOV{S}.W Rd, Rm, <shift> #imm
            sh = ((h2 >> 12) & 7) << 2 | (h2 >> 6) & 3
            typ, rd, val = (h2 >> 4) & 3, (h2 >> 8) & 15, r[h2 & 15]
            if typ == 0: res = (val << sh) & 0xFFFFFFFF
            elif typ == 1: res = val >> (sh or 32)
            else: raise Unsupported("shift type %d" % typ)
            r[rd] = res
            if h1 & 0x10: self.n, self.z = res >> 31, int(res == 0)
            t.append("mov.w")
        elif (h1 & 0xFFE0) == 0xEB00 and (h2 & 0x8000) == 0:         # ADD.W Rd, Rn, Rm{, LSL #imm}
            sh = ((h2 >> 12) & 7) << 2 | (h2 >> 6) & 3
            if (h2 >> 4) & 3: raise Unsupported("ADD.W shift type")
            r[(h2 >> 8) & 15] = (r[h1 & 15] + (r[h2 & 15] << sh)) & 0xFFFFFFFF; t.append("add.w")
        elif (h1 & 0xFBE0) == 0xF000 and (h2 & 0x8000) == 0:         # AND{S}.W Rd, Rn, #const
            imm = self._thumb_imm((h1 >> 10) & 1, (h2 >> 12) & 7, h2 & 0xFF)
            res = r[h1 & 15] & imm
            r[(h2 >> 8) & 15] = res
            if h1 & 0x10: self.n, self.z = res >> 31, int(res == 0)
            t.append("and.w")
        elif (h1 & 0xFBE0) == 0xF100 and (h2 & 0x8000) == 0:         # ADD{S}.W Rd, Rn, #const
            imm = self._thumb_imm((h1 >> 10) & 1, (h2 >> 12) & 7, h2 & 0xFF)
            rd = (h2 >> 8) & 15
            if h1 & 0x10: r[rd] = self._flags_add(r[h1 & 15], imm)
            else: r[rd] = (r[h1 & 15] + imm) & 0xFFFFFFFF
            t.append("add.w")
        elif (h1 & 0xFBF0) == 0xF1B0 and (h2 & 0x8F00) == 0x0F00:    # CMP.W Rn, #const
            self._flags_sub(r[h1 & 15], self._thumb_imm((h1 >> 10) & 1, (h2 >> 12) & 7, h2 & 0xFF)); t.append("cmp.w")
        elif (h1 & 0xF800) == 0xF000 and (h2 & 0xD000) == 0x8000:    # B<cond>.W
            S, cond = (h1 >> 10) & 1, (h1 >> 6) & 15
            J1, J2 = (h2 >> 13) & 1, (h2 >> 11) & 1
            off = (S << 20) | (J2 << 19) | (J1 << 18) | ((h1 & 0x3F) << 12) | ((h2 & 0x7FF) << 1)
            if S: off -= 1 << 21
            if self._cond(cond): self._branch(pc + 4 + off)
            t.append("bcond.w")
        elif (h1 & 0xF800) == 0xF000 and (h2 & 0xD000) in (0xD000, 0x9000):    # BL / B.W: a hooked external call, or the linked target
            hook = self.hooks.get(pc - self.code_base)
            if hook is not None:
                hook(self); t.append("bl")
            else:
                S, J1, J2 = (h1 >> 10) & 1, (h2 >> 13) & 1, (h2 >> 11) & 1
                I1, I2 = 1 - (J1 ^ S), 1 - (J2 ^ S)
                off = (S << 24) | (I1 << 23) | (I2 << 22) | ((h1 & 0x3FF) << 12) | ((h2 & 0x7FF) << 1)
                if S: off -= 1 << 25
                if (h2 & 0xD000) == 0xD000: r[14] = (pc + 4) | 1
                self._branch(pc + 4 + off); t.append("bl" if (h2 & 0xD000) == 0xD000 else "b.w")
        elif (h1 & 0xFFF0) in (0xF8B0, 0xF8D0, 0xF8C0, 0xF8A0):      # LDRH.W / LDR.W / STR.W / STRH.W Rt, [Rn, #imm12]
            addr, rt = r[h1 & 15] + (h2 & 0xFFF), (h2 >> 12) & 15
            if (h1 & 0xFFF0) == 0xF8B0: r[rt] = self.rd16(addr); t.append("ldrh.w")
            elif (h1 & 0xFFF0) == 0xF8D0:
                val = self.rd32(addr)
                if rt == 15: self._branch(val)
                else: r[rt] = val
                t.append("ldr.w")
            elif (h1 & 0xFFF0) == 0xF8C0: self.wr32(addr, r[rt]); t.append("str.w")
            else: struct.pack_into("<H", self.mem, addr, r[rt] & 0xFFFF); t.append("strh.w")
        elif (h1 & 0xFFF0) in (0xF830, 0xF850, 0xF840, 0xF820) and (h2 & 0x0800):   # ... [Rn, #-imm8] / pre- / post-indexed (T4)
            P, U, W = (h2 >> 10) & 1, (h2 >> 9) & 1, (h2 >> 8) & 1
            rn, rt, imm = h1 & 15, (h2 >> 12) & 15, h2 & 0xFF
            off_addr = r[rn] + (imm if U else -imm)
            addr = off_addr if P else r[rn]
            if (h1 & 0xFFF0) == 0xF830: r[rt] = self.rd16(addr); t.append("ldrh.w")
            elif (h1 & 0xFFF0) == 0xF850: r[rt] = self.rd32(addr); t.append("ldr.w")
            elif (h1 & 0xFFF0) == 0xF840: self.wr32(addr, r[rt]); t.append("str.w")
            else: struct.pack_into("<H", self.mem, addr, r[rt] & 0xFFFF); t.append("strh.w")
            if W: r[rn] = off_addr & 0xFFFFFFFF
        elif (h1 & 0xFBE0) == 0xF1A0 and (h2 & 0x8000) == 0:         # SUB{S}.W Rd, Rn, #const
            imm = self._thumb_imm((h1 >> 10) & 1, (h2 >> 12) & 7, h2 & 0xFF)
            rd = (h2 >> 8) & 15
            if h1 & 0x10: r[rd] = self._flags_sub(r[h1 & 15], imm)
            else: r[rd] = (r[h1 & 15] - imm) & 0xFFFFFFFF
            t.append("sub.w")
        elif (h1 & 0xFBEF) == 0xF04F and (h2 & 0x8000) == 0:         # MOV{S}.W Rd, #const
            imm = self._thumb_imm((h1 >> 10) & 1, (h2 >> 12) & 7, h2 & 0xFF)
            r[(h2 >> 8) & 15] = imm
            if h1 & 0x10: self.n, self.z = imm >> 31, int(imm == 0)
            t.append("mov.w")
        elif h1 == 0xFA1F and (h2 & 0xF0C0) == 0xF080:               # UXTH.W Rd, Rm{, ROR #}
            rot = ((h2 >> 4) & 3) * 8
            val = r[h2 & 15]
            r[(h2 >> 8) & 15] = ((val >> rot) | (val << (32 - rot))) & 0xFFFF; t.append("uxth")
        else:
            raise Unsupported("32-bit %04x %04x at +0x%x" % (h1, h2, pc - self.code_base))


def vfp_census(code):
    """Static count of the VFP data-processing instructions in a Thumb-2 code section (linear sweep; these functions hold no
    literal pools): {'vmul.f32': n, 'vadd.f32': n, 'vsub.f32': n, 'vmov.f32': n, 'fused': n, 'other': n}."""
    hw = struct.unpack("<%dH" % (len(code) // 2), code)
    out = {"vmul.f32": 0, "vadd.f32": 0, "vsub.f32": 0, "vnmul.f32": 0, "vmov.f32": 0, "fused": 0, "vdiv.f32": 0, "other": 0, "f64": 0}
    i = 0
    while i < len(hw):
        h1 = hw[i]
        if (h1 >> 11) not in (0b11101, 0b11110, 0b11111):
            i += 1
            continue
        h2 = hw[i + 1]
        i += 2
        if (h1 & 0xEF00) == 0xEE00 and (h2 & 0x0E10) == 0x0A00:
            if h2 & 0x0100:
                out["f64"] += 1
                continue
            opc1, op = (h1 >> 4) & 0xB, (h2 >> 6) & 1
            if opc1 in (0x0, 0x1, 0x9, 0xA): out["fused"] += 1          # vmla vmls vnmla vnmls vfnma vfnms vfma vfms
            elif opc1 == 0x2: out["vnmul.f32" if op else "vmul.f32"] += 1
            elif opc1 == 0x3: out["vsub.f32" if op else "vadd.f32"] += 1
            elif opc1 == 0x8: out["vdiv.f32"] += 1
            elif opc1 == 0xB and (h1 & 15) == 0 and (h2 >> 6) & 3 == 1: out["vmov.f32"] += 1
            else: out["other"] += 1
    return out


def load_function(archive, member, function):
    """-> (code bytes, [(offset, reloc type, symbol)]) of `.text.<function>` in `member` of `archive`."""
    elf = Elf32(ar_member(archive, member))
    if elf.e_machine != 40:
        raise ValueError("not an ARM object")
    sec = ".text." + function
    return elf.section(sec), elf.relocations(sec)


# ---------------------------------------------------------------------------------------------------------------------------
# a minimal static linker: sections of several archive members in one memory image, R_ARM_THM_CALL / _JUMP24 / _ABS32 resolved
class Image:
    """Places `.text.*` / `.rodata.*` sections of archive members in a Cpu's memory and resolves the relocations between them, so
    that a function can call the functions and read the tables it was compiled against (arm_cfft_f32 -> arm_cfft_radix8by2_f32 ->
    arm_radix8_butterfly_f32, arm_bitreversal_32; arm_cfft_sR_f32_len128 -> twiddleCoef_128, armBitRevIndexTable128)."""

    def __init__(self, cpu, base=0x1000):
        self.cpu, self.at, self.sym, self._pending = cpu, base, {}, []
        cpu.code_base = 0

    def add(self, archive, member, sections):
        elf = Elf32(ar_member(archive, member))
        placed = {}
        for sec in sections:
            body = elf.section(sec)
            self.at = (self.at + 15) & ~15
            self.cpu.mem[self.at:self.at + len(body)] = body
            placed[elf.names.index(sec)] = self.at
            self._pending.append((elf, sec, self.at))
            self.at += len(body)
        for name, value, size, info, shndx in elf.symbols():
            if name and shndx in placed and ((info & 15) in (1, 2) or (info >> 4) == 1):   # OBJECT / FUNC, or any global (assembly labels have no type)
                self.sym[name] = placed[shndx] + (value & ~1)
        return placed

    def link(self):
        cpu = self.cpu
        for elf, sec, at in self._pending:
            for off, rtype, symname in elf.relocations(sec):
                if symname not in self.sym:
                    raise KeyError("undefined symbol %s (needed by %s)" % (symname, sec))
                target, where = self.sym[symname], at + off
                if rtype in (10, 30):                                     # R_ARM_THM_CALL (BL), R_ARM_THM_JUMP24 (B.W): addend -4 in the instruction
                    rel = target - (where + 4)
                    S = 1 if rel < 0 else 0
                    rel &= (1 << 25) - 1
                    I1, I2 = (rel >> 23) & 1, (rel >> 22) & 1
                    J1, J2 = (1 - I1) ^ S, (1 - I2) ^ S
                    h1 = 0xF000 | (S << 10) | ((rel >> 12) & 0x3FF)
                    h2 = (0xD000 if rtype == 10 else 0x9000) | (J1 << 13) | (J2 << 11) | ((rel >> 1) & 0x7FF)
                    struct.pack_into("<HH", cpu.mem, where, h1, h2)
                elif rtype == 2:                                          # R_ARM_ABS32: word += symbol address
                    cpu.wr32(where, cpu.rd32(where) + target)
                else:
                    raise Unsupported("relocation type %d in %s" % (rtype, sec))
        self._pending = []
