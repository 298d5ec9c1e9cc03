#!/usr/bin/env python3
"""VGPR liveness of a kernel from its gfx950 assembly (hipcc -S --cuda-device-only -gline-tables-only).

Backward dataflow over the basic blocks of one function: per instruction the set of live VGPRs.  Prints the pressure
profile by source line (max live VGPRs while an instruction of that line executes) and the registers that stay live over
the longest stretches -- what a register diet has to look at.  Approximations: the first operand of an instruction is its
destination unless the mnemonic is a store / compare-to-SGPR / etc.; partial (dpp, sdwa, v_writelane) writes also read.

  python tools/isa_liveness.py /tmp/kg.s asdr_update_kernel
"""
import re
import sys
from collections import defaultdict

src, fn = sys.argv[1], sys.argv[2]
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith(fn + ":"))
end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start:end]

REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
NODEF = ("global_store", "scratch_store", "ds_write", "buffer_store", "s_", "v_cmp", "v_cmpx", "ds_append", "global_atomic")


def regs(tok):
    out = set()
    for m in REG.finditer(tok):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


insts = []      # (mnemonic, defs, uses, line)
labels = {}
cur_line = 0
for l in body:
    t = l.strip()
    m = re.match(r"\.loc\s+\d+\s+(\d+)", t)
    if m:
        cur_line = int(m.group(1)); continue
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        labels[m.group(1)] = len(insts); continue
    if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
        continue
    t = t.split(";")[0].strip()
    parts = t.split(None, 1)
    mn = parts[0]
    ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
    d, u = set(), set()
    if mn.startswith(NODEF):
        for o in ops: u |= regs(o)
        if mn.startswith("v_cmp") and ops and ops[0].startswith("v"):   # v_cmp_* e64 to SGPR: no VGPR def
            pass
    else:
        if ops:
            d = regs(ops[0])
        for o in ops[1:]: u |= regs(o)
        if "dpp" in t or "sdwa" in t or mn.startswith("v_writelane") or mn.startswith("v_fmac") or mn.startswith("v_mac") or mn.startswith("v_pk_fma"):
            u |= d
    tgt = None
    if mn.startswith("s_cbranch") or mn == "s_branch":
        tgt = ops[0] if ops else None
    insts.append([mn, d, u, cur_line, tgt])

n = len(insts)
succ = [[] for _ in range(n)]
for i, (mn, d, u, ln, tgt) in enumerate(insts):
    if mn == "s_branch":
        if tgt in labels: succ[i].append(labels[tgt])
    else:
        if i + 1 < n and mn != "s_endpgm": succ[i].append(i + 1)
        if tgt in labels: succ[i].append(labels[tgt])
live_in = [set() for _ in range(n)]
changed = True
it = 0
while changed and it < 200:
    changed = False; it += 1
    for i in range(n - 1, -1, -1):
        out = set()
        for s_ in succ[i]: out |= live_in[s_]
        new = (out - insts[i][1]) | insts[i][2]
        if new != live_in[i]:
            live_in[i] = new; changed = True
press = [len(s_) for s_ in live_in]
print("instructions %d, dataflow iterations %d, max live VGPRs %d" % (n, it, max(press)))
by_line = defaultdict(int)
for i in range(n): by_line[insts[i][3]] = max(by_line[insts[i][3]], press[i])
print("source lines with the highest pressure:")
for ln, p in sorted(by_line.items(), key=lambda kv: -kv[1])[:25]:
    print("  line %5d  %3d live" % (ln, p))
# long-lived registers: fraction of instructions at which each VGPR is live
cnt = defaultdict(int)
for s_ in live_in:
    for r in s_: cnt[r] += 1
print("registers live over the largest share of the kernel:")
for r, c in sorted(cnt.items(), key=lambda kv: -kv[1])[:48]:
    print("  v%-3d %5.1f %%" % (r, 100.0 * c / n), end="")
    if (sorted(cnt.items(), key=lambda kv: -kv[1]).index((r, c)) + 1) % 6 == 0: print()
print()
# pressure along the program (coarse): max per 200-instruction window with the dominant source line
print("pressure along the program:")
for w in range(0, n, 250):
    seg = range(w, min(n, w + 250))
    p = max(press[i] for i in seg)
    ls = defaultdict(int)
    for i in seg: ls[insts[i][3]] += 1
    top = max(ls.items(), key=lambda kv: kv[1])[0]
    print("  inst %5d..%5d  max %3d  min %3d  (mostly line %d)" % (w, w + 249, p, min(press[i] for i in seg), top))
