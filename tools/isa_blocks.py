#!/usr/bin/env python3
"""Basic blocks of one kernel in hipcc's assembly output (`hipcc -S --cuda-device-only -gline-tables-only`):
per block the instruction counts by class (VALU / of which v_mov, v_cndmask, v_pk_*, f64 / SALU / LDS / VMEM / waits), the
source lines it comes from and whether it ends in a backward branch (a loop body).  Multiply by the trip counts by hand:
the kernel is issue-bound, so instructions per wave is the number that matters.

    python tools/isa_blocks.py /tmp/isa/base.s asdr_update_kernel [min_instructions]
"""
import collections
import re
import sys


def main():
    path, kern = sys.argv[1], sys.argv[2]
    min_n = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(kern + ":"))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    blocks, cur, order = collections.OrderedDict(), "entry", {}
    blocks[cur] = []
    src = None
    for l in lines[start + 1:end + 1]:
        s = l.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            cur = m.group(1); blocks[cur] = []; order[cur] = len(order); continue
        m = re.match(r"^\.loc\s+\d+\s+(\d+)", s)
        if m:
            src = int(m.group(1)); continue
        if not s or s.startswith((".", ";", "//")) or s.endswith(":"):
            continue
        blocks[cur].append((s.split()[0], s, src))
    tot = collections.Counter()
    print("%-12s %5s %5s %4s %4s %4s %4s %5s %4s %4s %4s  %s" % ("block", "all", "valu", "mov", "cnd", "pk", "f64", "salu", "lds", "vmem", "wait", "source lines (most frequent) / loop"))
    for name, ins in blocks.items():
        c = collections.Counter()
        srcs = collections.Counter()
        loop = ""
        for op, s, ln in ins:
            srcs[ln] += 1
            if op.startswith("v_"):
                c["valu"] += 1
                if op.startswith("v_mov") or op.startswith("v_accvgpr"): c["mov"] += 1
                if op.startswith("v_cndmask"): c["cnd"] += 1
                if op.startswith("v_pk_"): c["pk"] += 1
                if "f64" in op: c["f64"] += 1
            elif op.startswith("s_waitcnt") or op.startswith("s_nop"):
                c["wait"] += 1
            elif op.startswith("s_"):
                c["salu"] += 1
                m = re.search(r"s_cbranch\w*\s+(\.LBB\d+_\d+)", s)
                if m and m.group(1) in order and (name == m.group(1) or order.get(m.group(1), 1 << 30) <= order.get(name, -1)):
                    loop = "  <- loops to " + m.group(1)
            elif op.startswith("ds_"):
                c["lds"] += 1
            elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
                c["vmem"] += 1
        n = len(ins)
        for k, v in c.items(): tot[k] += v
        tot["all"] += n
        if n >= min_n or loop:
            top = ",".join("%s" % k for k, _ in srcs.most_common(4))
            print("%-12s %5d %5d %4d %4d %4d %4d %5d %4d %4d %4d  %s%s" % (name, n, c["valu"], c["mov"], c["cnd"], c["pk"], c["f64"], c["salu"], c["lds"], c["vmem"], c["wait"], top, loop))
    print("static total:", dict(tot))


if __name__ == "__main__":
    main()
